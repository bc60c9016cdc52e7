#!/bin/bash
# A/B of alignment-kernel variants: duration of the FIRST k_alnf<0,14> launch of each run (round 0: the same 1.7 M jobs whatever the library
# does with the results).  usage (GPU box, repo root): bash tools/first_aln_ab.sh libtksmseq.so libtksmseq_<variant>.so ...   (names under tksm_amd/)
R=$PWD
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
for lib in "$@"; do
  rm -rf $R/gpurun_out/fa
  TKSMSEQ_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fa -- python3 $R/tools/quick_stage_times.py 1703936 > $R/gpurun_out/fa.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/fa/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]; seen_init=False
for r in rows:
    k=r["Kernel_Name"]
    if "k_init" in k: seen_init=True
    if seen_init and "k_alnf<0, 14" in k:
        out.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6); seen_init=False
print("$lib first 14-row launch of each run (ms):", [round(x,2) for x in out])
PY
done
rm -rf $R/gpurun_out/fa
