#!/bin/bash
# kernel trace + stats of one context running alone: bash tools/trace_qst.sh <tag> [reads] [env assignments...]
tag=$1; B=${2:-1310720}; shift; shift
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
# the queue configuration of the headline run (bench.py / the CLI set it themselves, but under rocprofv3 --pmc the runtime starts before the program does)
export GPU_MAX_HW_QUEUES=16
TKSMSEQ_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/tools/quick_stage_times.py $B > $out/trace.log 2>&1
cd $R
cp $out/trace/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
grep -h "wall\|reads .* rounds" $out/trace.log
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/kernel_stats.csv')))
for r in rows:
    n=r['Name'].split('(')[0].replace('tk::','').replace('void ','')
    ms=float(r['TotalDurationNs'])/1e6/3
    if ms>0.3 and n.startswith('k_'): print(f"{n:28s} calls/run {int(r['Calls'])/3:6.1f}  ms/run {ms:7.2f}  avg us {float(r['AverageNs'])/1e3:8.1f}")
PY
