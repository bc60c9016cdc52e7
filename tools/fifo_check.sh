#!/bin/bash
# diagnostic: `tksm sequence` writing to a pipe (the ordered writer thread)
R=$PWD
d=$R/tests/golden/splice_corpus
export TKSM_MODELS=$R/tksm_amd/models
rm -f /tmp/pipe.fastq /tmp/fifo_copy.fastq; mkfifo /tmp/pipe.fastq
cat /tmp/pipe.fastq > /tmp/fifo_copy.fastq &
timeout -k 5 60 $R/tksm_amd/tksm sequence -i $d/mols.mdf -r $d/ref.fa -s 11 --batch-bytes 4096 -o /tmp/pipe.fastq --devices 0 --in-flight 3 --verbosity DEBUG
echo rc=$?
wait
timeout -k 5 60 $R/tksm_amd/tksm sequence -i $d/mols.mdf -r $d/ref.fa -s 11 --batch-bytes 4096 -o /tmp/direct.fastq --devices 0 --in-flight 1 --verbosity OFF
cmp /tmp/fifo_copy.fastq /tmp/direct.fastq && echo same bytes
