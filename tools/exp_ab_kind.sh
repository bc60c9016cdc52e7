#!/bin/bash
# A/B of builds on another workload kind: bash tools/exp_ab_kind.sh <kind> <rounds> <lib> <lib> ...
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
K=$1; N=$2; shift; shift
out=gpurun_out/exp_ab_kind.log
: > $out
B="python bench.py --kind $K --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --no-side-legs"
one() { echo -n "$K $1: " >> $out; TKSMSEQ_LIB=$1 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
x=d.get('roofline',{}).get('exclusive_ms_per_step') or {}
print(round(d['value']/1e6,3), 'M reads/s', round(d['ms_per_step'],2), 'ms/step; exclusive', {k: round(v,2) for k,v in x.items()})" >> $out; }
for i in $(seq $N); do for l in "$@"; do one $l; done; done
cat $out
