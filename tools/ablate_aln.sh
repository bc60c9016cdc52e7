#!/bin/bash
# k_aln launch durations of the first rounds with parts of the kernel switched off (diagnostic build: make ablate)
set -e
R=$PWD
export TKSMSEQ_LIB=$R/tksm_amd/libtksmseq_prof.so
cd /tmp && export TMPDIR=/tmp
for a in "$@"; do
  export TKSMSEQ_ABLATE=$a
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab$a -- python $R/tools/quick_stage_times.py 1310720 > $R/gpurun_out/ab$a.log 2>&1 || true
  echo ablate $a
  python $R/tools/first_aln.py $R/gpurun_out/ab$a
  rm -rf $R/gpurun_out/ab$a
done
