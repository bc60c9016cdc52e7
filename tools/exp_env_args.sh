#!/bin/bash
# A/B of environment knobs with extra bench arguments: BENCH_ARGS="..." bash tools/exp_env_args.sh "VAR=a" "VAR=b" ...
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/exp_env.log
: > $out
B="python bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-e2e --no-side-legs $BENCH_ARGS"
for e in "$@"; do
  echo -n "$e: " >> $out
  env $e $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
x=d.get('roofline',{}).get('exclusive_ms_per_step') or {}
print(round(d['value']/1e6,3), 'M reads/s', round(d.get('gbases_per_s',0),2), 'Gb/s', round(d['ms_per_step'],2), 'ms/step; exclusive', {k: round(v,2) for k,v in x.items()})" >> $out
done
cat $out
