#!/bin/bash
# Usage (GPU box): tools/sweep.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   - rebuild trace.hip with each build-time setting and bench it
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip editable-gaussian-reflections_amd/csrc/bvh.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$TAG/build_$NAME.log; continue; }
  env $CFG python bench.py --no-cpu-baseline --steps 60 --warmup 40 ${SWEEP_ARGS:---no-second-variant} > gpurun_out/$TAG/bench_$NAME.json 2> gpurun_out/$TAG/bench_$NAME.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_$NAME.json")); o=d.get("other_variant") or {}
    print("$CFG:", d["value"], d["ms_per_step"], {k:d["kernel_ms"][k] for k in ("forward_chain","backward_chain")}, "status", d["status"], "| other", o.get("value"), (o.get("kernel_ms") or {}).get("forward_chain"), (o.get("kernel_ms") or {}).get("backward_chain"))
except Exception as e:
    print("$CFG: FAILED", e)
PY
done
