#!/bin/bash
# second soak of round 4: other seeds, models, identity settings, short and heavy-tailed lengths (tools/soak_parity.py; GPU box)
set -e
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/soak_r04b.log
: > $out
run() { echo "## $*" >> $out; "$@" 2>&1 | grep -E "^(oracle:|gpu:|RESULT)" >> $out; }
SOAK_MODEL=nanopore2018 SOAK_IDENT=90,98,4 run python tools/soak_parity.py 300000 scrna 1000 0 0 21
SOAK_MODEL=pacbio2016 SOAK_IDENT=84,99,5.5 run python tools/soak_parity.py 300000 pcr 800 0 0 22
run python tools/soak_parity.py 400000 bulk 300 0 0 23
run python tools/soak_parity.py 100000 bulk 2500 0.8 0 24
SOAK_IDENT=60,80,10 run python tools/soak_parity.py 120000 bulk 1000 0 0 25
SOAK_IDENT=90,90,0 run python tools/soak_parity.py 200000 bulk 1000 0 0 26
run python tools/soak_parity.py 150000 scrna 1000 0 1 27
echo "soak done" >> $out
