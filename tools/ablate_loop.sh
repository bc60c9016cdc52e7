#!/bin/bash
# k_loop launch durations with parts of its first phase switched off (diagnostic build: make ablate; results are wrong, times are not):
# 34 no first-level threshold gather, 35 every k-mer from LDS, 36 both.  usage (GPU box, repo root): bash tools/ablate_loop.sh 0 34 35 36
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
    if 'k_loop' in r['Name']: print('ablate $a:', r['Name'][:16], r['Calls'], 'launches', round(int(r['TotalDurationNs'])/3e6, 2), 'ms per step')"
  rm -rf $R/gpurun_out/ab$a
done
