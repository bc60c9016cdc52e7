#!/bin/bash
# A/B of queue priorities (gpurun box): the wide alignment passes on a stream of lower priority than everything else
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
out=gpurun_out/exp_prio.log
: > $out
python - >> $out 2>&1 <<'PY'
import torch
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range())
PY
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e --no-side-legs"
one() { echo "## $1" >> $out; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']/1e6,3), 'M reads/s', round(d['ms_per_step'],2), 'ms/step')" >> $out; }
one "baseline" X=1
one "alnf high (-1), main normal" TKSMSEQ_ALN_STREAM_PRIORITY=-1
one "baseline" X=1
one "alnf high (-1), main normal" TKSMSEQ_ALN_STREAM_PRIORITY=-1
one "baseline" X=1
one "alnf high (-1), main normal" TKSMSEQ_ALN_STREAM_PRIORITY=-1
cat $out
