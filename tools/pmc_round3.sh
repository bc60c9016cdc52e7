#!/bin/bash
# Round-3 diagnostics: SQ / TCC counters per kernel of one context running alone (tools/quick_stage_times.py), in separate passes.
# usage (GPU box, repo root): bash tools/pmc_round3.sh <tag> [reads]
tag=$1; B=${2:-1310720}
R=$PWD; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the queue configuration of the headline run (bench.py / the CLI set it themselves, but under rocprofv3 --pmc the runtime starts before the program does)
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/tools/quick_stage_times.py $B > $out/trace.log 2>&1 && echo trace >> $out/progress.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $out/sq -- python $R/tools/quick_stage_times.py $B > $out/sq.log 2>&1 && echo sq >> $out/progress.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/sq2 -- python $R/tools/quick_stage_times.py $B > $out/sq2.log 2>&1 && echo sq2 >> $out/progress.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/tcc -- python $R/tools/quick_stage_times.py $B > $out/tcc.log 2>&1 && echo tcc >> $out/progress.log
cd $R
for p in sq sq2 tcc; do python tools/pmc_kernels.py $out/$p > $out/$p.txt 2>&1; done
cp $out/trace/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace $out/sq $out/sq2 $out/tcc
tail -4 $out/trace.log
