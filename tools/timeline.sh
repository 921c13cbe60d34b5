#!/bin/bash
# Usage (GPU box): tools/timeline.sh <tag>  - kernel start/end timeline of the last iteration (rocprofv3 --kernel-trace)
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ITERS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG -o t -- python tools/pmc_run.py > gpurun_out/$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last iteration = after the last k_prologue
idx=max(i for i,r in enumerate(rows) if "k_prologue" in r["Kernel_Name"])
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n=r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::","").replace("void ","")
    print(f'{n[:28]:28s} q={r.get("Queue_Id","?"):>3s} grid={r.get("Grid_Size_X", r.get("Grid_Size","?")):>8s} start={(int(r["Start_Timestamp"])-t0)/1e6:8.3f} end={(int(r["End_Timestamp"])-t0)/1e6:8.3f} ms')
PY
