#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
tag=$1
R=$PWD
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the queue configuration of the headline run (bench.py / the CLI set it themselves, but under rocprofv3 --pmc the runtime starts before the program does)
export GPU_MAX_HW_QUEUES=16
python $R/bench.py --no-e2e > $out/bench.json 2> $out/bench.err
echo bench done >> $out/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline --no-e2e --no-side-legs > $out/bench_under_rocprof.json 2> $out/stats.err
echo stats done >> $out/progress.log
# one context at a time: every kernel has the GPU to itself, so total duration / steps is the exclusive time bench.py reports as kernel_ms
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -- python $R/bench.py --no-cpu-baseline --no-e2e --no-side-legs --pipeline 1 --steps 4 > $out/bench_single_ctx_under_rocprof.json 2> $out/stats1.err
echo single-context stats done >> $out/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-side-legs > $out/fetch.log 2>&1
echo fetch done >> $out/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-side-legs > $out/write.log 2>&1
echo write done >> $out/progress.log
cd $R
python tools/make_traffic_json.py $out 7 1703936 $tag > $out/hbm_traffic.json   # 3 warm-up + 3 timed steps + the exclusive step (the pmc passes run without side legs)
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
cp $out/stats1/*/*kernel_stats.csv $out/single_ctx_kernel_stats.csv
rm -rf $out/stats $out/stats1 $out/fetch/*/*agent_info.csv
ls -la $out
cat $out/bench.json
