#!/bin/bash
# Usage (GPU box): tools/timeline.sh <tag>  - kernel start/end timeline of the last iteration (rocprofv3 --kernel-trace)
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ITERS=6 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -o t -- python tools/pmc_run.py > gpurun_out/$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
pro=[i for i,r in enumerate(rows) if "k_prologue" in r["Kernel_Name"]]
a,b=pro[-2],pro[-1]   # one full iteration: prologue to the next prologue
it=rows[a:b]
t0=int(it[0]["Start_Timestamp"]); span=(int(rows[b]["Start_Timestamp"])-t0)/1e6
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in it)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print(f"iteration span {span:.3f} ms, GPU busy (union of kernels) {busy/1e6:.3f} ms, idle {span-busy/1e6:.3f} ms, kernels {len(it)}")
for r in it:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    print(f'{n[:34]:34s} q={r.get("Queue_Id","?"):>3s} start={(int(r["Start_Timestamp"])-t0)/1e6:8.3f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6:7.3f} ms')
PY
