#!/bin/bash
# A/B of CLI variants on ONE box: generates the 8 M-molecule input once, then runs `tksm sequence` with each TKSMSEQ_RAMP value in turn, three rounds
cd $GRAFT_REPO_ROOT
E2E_MODES=none python tools/e2e_cli.py 8000000 > /dev/null 2>&1
d=/tmp/e2e
export TKSM_MODELS=$PWD/tksm_amd/models
for round in 1 2 3; do
  for r in "$@"; do
    rm -f $d/b.fastq
    s=$(date +%s%N)
    TKSMSEQ_RAMP=$r tksm_amd/tksm sequence -i $d/mols.mdf -r $d/ref.fa -o $d/b.fastq --verbosity ERROR > /dev/null 2>&1
    e=$(date +%s%N)
    echo "ramp $r: $(( (e - s) / 1000000 )) ms"
  done
done
