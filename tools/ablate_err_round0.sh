#!/bin/bash
# diagnostic: duration of the round-0 k_err launches for a given diagnostic library (k_aln disabled: the run ends after a few rounds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  rm -rf gpurun_out/abl
  TKSMSEQ_ABLATE=10 TKSMSEQ_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl -- python tools/quick_stage_times.py 1048576 > gpurun_out/abl.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "tk::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=max(i for i,r in enumerate(rows) if "k_init" in r["Kernel_Name"])
t=0.0
for r in rows[idx+1:]:
    if "k_err" not in r["Kernel_Name"]: break
    t+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
print("$lib round-0 k_err %.2f ms"%t)
PY
done
