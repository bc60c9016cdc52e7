#!/bin/bash
# diagnostic: time of the first round's k_err launches when the kernel returns early at point $1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in ${ABL:-1 6 2 3 4 0}; do
  rm -rf gpurun_out/abl
  TKSMSEQ_ABLATE=$a TKSMSEQ_TAIL_CUT=0 TKSMSEQ_LIB=libtksmseq_prof.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -- python tools/quick_stage_times.py 1048576 > gpurun_out/abl.log 2>&1; echo run $a done >> gpurun_out/abl_progress.log
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "k_err" in r["Kernel_Name"] or "k_init" in r["Kernel_Name"] or "k_aln" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=max(i for i,r in enumerate(rows) if "k_init" in r["Kernel_Name"])
tot=0
for r in rows[idx+1:]:
    if "k_err" not in r["Kernel_Name"]: break
    tot+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
print("ablate=$a first-round k_err total %.2f ms"%(tot/1e6))
PY
done
