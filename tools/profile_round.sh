#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -e
tag=$1
R=$PWD
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $out/bench.json 2> $out/bench.err
echo bench done >> $out/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
echo stats done >> $out/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
echo fetch done >> $out/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
echo write done >> $out/progress.log
cd $R
python tools/make_traffic_json.py $out 6 1310720 > $out/hbm_traffic.json   # 3 timed steps + 1 warm-up step per context
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/stats $out/fetch/*/*agent_info.csv
ls -la $out
cat $out/bench.json
