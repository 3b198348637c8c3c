#!/bin/bash
# timeline of the frame pipelines: kernel + memory-copy trace of a short e2e run (rocprofv3), CSVs under gpurun_out/e2e_trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/e2e_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/e2e_trace -- python $R/tools/e2e_pipeline.py --frames ${1:-12} ${@:2} > $R/gpurun_out/e2e_trace.log 2>&1
tail -3 $R/gpurun_out/e2e_trace.log
find $R/gpurun_out/e2e_trace -name "*.csv" | head
python - <<'PY'
import csv, glob, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
ev=[]
for f in glob.glob(R+"/gpurun_out/e2e_trace/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Direction"] if "Direction" in r else r.get("Name","copy"), r))
for f in glob.glob(R+"/gpurun_out/e2e_trace/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K:"+r["Kernel_Name"][:40], r))
ev.sort()
if ev:
    t0=ev[0][0]
    big=[e for e in ev if e[1]-e[0]>200000 or e[2].startswith("K:assemble")]
    out=open(R+"/gpurun_out/e2e_timeline.txt","w")
    for s,e,n,r in big[-400:]:
        out.write("%10.3f %10.3f %8.3f ms  %s %s\n"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n,r.get("Size",r.get("Bytes",""))))
    out.close()
    print("events",len(ev),"big",len(big))
PY
