#!/bin/bash
# Usage (GPU box): tools/pmc_quick.sh <tag> COUNTER [COUNTER...]   - one time-boxed rocprofv3 --pmc pass per counter on one iteration
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for C in "$@"; do
  C=${C//,/ }
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/$TAG/${C// /_} -o p -- python tools/pmc_run.py > "gpurun_out/$TAG.${C// /_}.log" 2>&1
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/$TAG/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:44]+"#"+r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:int(kv[0].split("#")[1])):
    if any(s in k for s in ("k_b","k_forward","k_step")): print(k, {a:round(b/1e6,2) for a,b in v.items()})
PY
