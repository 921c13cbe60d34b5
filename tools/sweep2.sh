#!/bin/bash
# Usage (GPU box): tools/sweep2.sh <tag> "ENV1=a ENV2=b" ...  - like sweep.sh, with the primary-only leg and the per-kernel times of both variants on one line each
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$TAG/build_$NAME.log; continue; }
  env $CFG python bench.py --no-cpu-baseline --steps 60 --warmup 40 ${SWEEP_ARGS} > gpurun_out/$TAG/bench_$NAME.json 2> gpurun_out/$TAG/bench_$NAME.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_$NAME.json")); o=d.get("other_variant") or {}
    km=lambda k: {x:k[x] for x in ("forward_chain","backward_chain") if x in k}
    print("$CFG: init", d["value"], d["ms_per_step"], km(d["kernel_ms"]), "primary-only", (d.get("primary_only") or {}).get("ms_per_step"), "| trained", o.get("value"), o.get("ms_per_step"), km(o.get("kernel_ms") or {}), "primary-only", (o.get("primary_only") or {}).get("ms_per_step"), "status", d["status"])
except Exception as e:
    print("$CFG: FAILED", e)
PY
done
