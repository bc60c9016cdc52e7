#!/bin/bash
# diagnostic: duration of the first k_aln launch when the kernel returns early (10: after meta, 11: after the forward pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in ${ABL:-10 11 0}; do
  rm -rf gpurun_out/abl
  TKSMSEQ_ABLATE=$a TKSMSEQ_TAIL_CUT=100000000 TKSMSEQ_LIB=libtksmseq_prof.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -- python tools/quick_stage_times.py 1048576 > gpurun_out/abl.log 2>&1
  echo run $a done >> gpurun_out/abl_progress.log
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "k_aln" in r["Kernel_Name"] or "k_init" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=max(i for i,r in enumerate(rows) if "k_init" in r["Kernel_Name"])
r=rows[idx+1]
print("ablate=$a first k_aln %.2f ms"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
done
