#!/bin/bash
# A/B of environment knobs of the library on one box: bash tools/exp_env.sh "VAR=a" "VAR=b" ...   (each argument: one run's extra environment; "X=1" = baseline)
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/exp_env.log
: > $out
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --no-side-legs"
for e in "$@"; do
  echo -n "$e: " >> $out
  env $e $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
x=d.get('roofline',{}).get('exclusive_ms_per_step') or {}
print(round(d['value']/1e6,3), 'M reads/s', round(d['ms_per_step'],2), 'ms/step; exclusive', {k: round(v,2) for k,v in x.items()})" >> $out
done
cat $out
