#!/bin/bash
# k_err (last visit) launch durations with parts of its q-score loop switched off (diagnostic build: make ablate)
set -e
R=$PWD
export TKSMSEQ_LIB=$R/tksm_amd/libtksmseq_prof.so
cd /tmp && export TMPDIR=/tmp
# the queue configuration of the headline run (bench.py / the CLI set it themselves, but under rocprofv3 --pmc the runtime starts before the program does)
export GPU_MAX_HW_QUEUES=16
for a in "$@"; do
  export TKSMSEQ_ABLATE=$a
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab$a -- python $R/tools/quick_stage_times.py 1310720 > $R/gpurun_out/ab$a.log 2>&1 || true
  python -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$R/gpurun_out/ab$a/*/*kernel_stats.csv')[0])):
    if 'k_err' in r['Name']: print('ablate $a: k_err', r['Calls'], 'launches', round(int(r['TotalDurationNs'])/3e6, 2), 'ms per step')"
  rm -rf $R/gpurun_out/ab$a
done
