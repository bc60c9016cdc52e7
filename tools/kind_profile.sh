#!/bin/bash
# exclusive per-kernel time of one workload kind (one context alone): bash tools/kind_profile.sh <kind> [extra bench args]   (GPU box)
kind=$1; shift
R=$PWD; out=$R/gpurun_out/kind_$kind; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --kind $kind --no-cpu-baseline --no-e2e --no-side-legs --pipeline 1 --steps 4 "$@" > $out/bench.json 2> $out/err.log
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv; rm -rf $out/stats
python3 - <<PY
import csv,json
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
d=json.loads(open("$out/bench.json").readline())
steps=d['steps']+d['warmup']+1
print("$kind", round(d['value']/1e6,2), 'M reads/s single context; steps under the profiler', steps)
for r in rows[:11]:
    if 'tk::' in r['Name']: print('  ', r['Name'][:46], r['Calls'], round(int(r['TotalDurationNs'])/1e6/steps,2), 'ms per step')
PY
