#!/bin/bash
# Usage (GPU box): tools/sweep_emu3.sh <tag> <world> "BUILDENV1" "BUILDENV2" ... - rank 0 of an emulated partition (team help on) per build-time setting, both clouds, two runs each
TAG=$1; W=$2; shift 2
mkdir -p gpurun_out/$TAG
for B in "$@"; do
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $B python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build.log 2>&1 || { echo "$B: BUILD FAILED"; tail -5 gpurun_out/$TAG/build.log; continue; }
  for V in init trained; do for rep in 1 2; do
    env $B python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $W --variant $V 2>/dev/null | tail -1 > gpurun_out/$TAG/emu.json
    python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/emu.json"))
print("$B | world $W $V:", d["ms_per_step"], {k: d["kernel_ms"][k] for k in ("forward_chain", "backward_chain")}, "status", d["status"])
PY
  done; done
done
