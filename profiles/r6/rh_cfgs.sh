#!/bin/bash
# Usage (GPU box): tools/rh_cfgs.sh <tag> "<build env>" ... - per build setting: remote help off / on / off / on, forward chain ms of both clouds + the protocol's statistics of the last launch
T=$1; shift
mkdir -p gpurun_out/$T
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$T/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$T/build_$NAME.log; continue; }
  for rh in 0 1 0 1; do
    EGR_PRINT_RH_STATS=1 EGR_REMOTE_HELP=$rh timeout 300 python bench.py --no-cpu-baseline ${RH_ARGS} > gpurun_out/$T/b.json 2> gpurun_out/$T/b.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$T/b.json")); o=d.get("other_variant") or {}
    print("$CFG rh=$rh: init fwd", d["kernel_ms"]["forward_chain"], "status", d["status"], "| trained fwd", (o.get("kernel_ms") or {}).get("forward_chain"), "status", o.get("status"))
except Exception as e:
    print("$CFG rh=$rh FAILED", e); print(open("gpurun_out/$T/b.err").read()[-800:])
PY
  done
  grep "remote help" gpurun_out/$T/b.err | grep -v "offers 0" | head -2 | cut -c1-420
done
