#!/bin/bash
# A/B of two builds of the library on one box, alternating: bash tools/exp_ab.sh <name under tksm_amd/> <name> [rounds=3]
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
A=$1; Bn=$2; N=${3:-3}
out=gpurun_out/exp_ab.log
: > $out
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --no-side-legs"
one() { echo -n "$1: " >> $out; TKSMSEQ_LIB=$1 $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
x=d.get('roofline',{}).get('exclusive_ms_per_step') or {}
print(round(d['value']/1e6,3), 'M reads/s', round(d['ms_per_step'],2), 'ms/step; exclusive', {k: round(v,2) for k,v in x.items()} if isinstance(x,dict) else '')" >> $out; }
for i in $(seq $N); do one $A; one $Bn; done
cat $out
