#!/bin/bash
# Usage (GPU box): tools/sweep_emu.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...  - rebuild with each build-time setting; bench the whole image (both clouds) and rank 0 of an emulated 8-way partition
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$TAG/build_$NAME.log; continue; }
  env $CFG python bench.py --no-cpu-baseline --steps 60 --warmup 40 --primary-steps 0 > gpurun_out/$TAG/bench_$NAME.json 2> gpurun_out/$TAG/bench_$NAME.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_$NAME.json")); o=d.get("other_variant") or {}
    print("$CFG: whole image init/trained", d["value"], o.get("value"), "fwd", d["kernel_ms"]["forward_chain"], (o.get("kernel_ms") or {}).get("forward_chain"), "bwd", d["kernel_ms"]["backward_chain"], (o.get("kernel_ms") or {}).get("backward_chain"), "status", d["status"], o.get("status"))
except Exception as e:
    print("$CFG: FAILED", e)
PY
  for V in init trained; do
    env $CFG python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world 8 --variant $V ${EMU_ARGS} 2>/dev/null | tail -1 > gpurun_out/$TAG/emu_$NAME_$V.json
    python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/emu_$NAME_$V.json"))
print("   emu8 $V:", d["ms_per_step"], d["kernel_ms"])
PY
  done
done
