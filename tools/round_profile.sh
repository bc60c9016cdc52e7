#!/bin/bash
# per-round kernel durations of one 2M-read step (rocprofv3 kernel trace); usage: bash tools/round_profile.sh [B]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16     # the queue configuration of the headline run
B=${1:-2097152}
rm -rf gpurun_out/rp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rp -- python tools/quick_stage_times.py $B > gpurun_out/rp.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/rp/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "tk::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: from the last k_init on
idx=max(i for i,r in enumerate(rows) if "k_init" in r["Kernel_Name"])
rows=rows[idx:]
t0=int(rows[0]["Start_Timestamp"])
rnd=-1; cur=None; out=[]
for r in rows:
    k=r["Kernel_Name"].split("tk::")[1].split("(")[0].split("<")[0]
    s=(int(r["Start_Timestamp"])-t0)/1e6; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if k=="k_err":
        if cur is None or cur["aln"]>0 or cur.get("closed"):
            cur={"start":s,"err":0.0,"aln":0.0,"nerr":0}; out.append(cur)
        cur["err"]+=d; cur["nerr"]+=1; cur["end"]=s+d
    elif k=="k_aln":
        cur["aln"]+=d; cur["end"]=s+d
    else:
        print("%-22s start %8.2f dur %7.3f"%(k,s,d)); 
        if cur: cur["closed"]=True
prev=None
for i,c in enumerate(out):
    gap=c["start"]-prev if prev is not None else 0.0
    print("round %2d start %8.2f gap %6.3f err %7.3f (%2d launches) aln %7.3f"%(i,c["start"],gap,c["err"],c["nerr"],c["aln"]))
    prev=c["end"]
PY
