#!/bin/bash
# A/B of CLI variants on ONE box: generates the 8 M-molecule input once, then runs `tksm sequence` with each value of $E2E_VAR (an environment
# variable of the build under test) in turn, three rounds, E2E_SLEEP seconds apart.  The pause matters: a run that starts right after another
# process released ~100 GiB of device memory takes 1.5 - 2 x as long (1.88 s -> 3.5 s to /dev/null), and a file run also competes with the
# write-back of the previous run's 16 GB -- a first experiment with smaller first batches looked 8 % faster without pauses and was 5 - 20 %
# slower with them (and is gone again).
cd $GRAFT_REPO_ROOT
E2E_MODES=none python tools/e2e_cli.py 8000000 > /dev/null 2>&1
d=/tmp/e2e
out=${E2E_OUT:-$d/b.fastq}      # E2E_OUT=null: a symlink to /dev/null (the ordered-writer path)
if [ "$out" = null ]; then ln -sf /dev/null $d/null.fastq; out=$d/null.fastq; fi
export TKSM_MODELS=$PWD/tksm_amd/models
for round in 1 2 3; do
  for r in "$@"; do
    [ -L $out ] || rm -f $out
    s=$(date +%s%N)
    env ${E2E_VAR:-TKSMSEQ_UNUSED}=$r tksm_amd/tksm sequence -i $d/mols.mdf -r $d/ref.fa -o $out --verbosity ERROR > /dev/null 2>&1
    e=$(date +%s%N)
    echo "${E2E_VAR:-variant} $r: $(( (e - s) / 1000000 )) ms"; sleep ${E2E_SLEEP:-8}
  done
done
