#!/bin/bash
# Usage (GPU box): tools/sweep_emu2.sh <tag> <world> "BUILDENV" -- "RUNENV1" "RUNENV2" ...  - one build, rank 0 of an emulated partition under several run-time settings
TAG=$1; W=$2; B=$3; shift 4
mkdir -p gpurun_out/$TAG
touch editable-gaussian-reflections_amd/csrc/trace.hip
env $B python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build.log 2>&1 || { echo "BUILD FAILED"; tail -5 gpurun_out/$TAG/build.log; exit 1; }
for R in "$@"; do for V in init trained; do
    env $B $R python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $W --variant $V 2>/dev/null | tail -1 > gpurun_out/$TAG/emu.json
    python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/emu.json"))
print("$B | $R | world $W $V:", d["ms_per_step"], {k: d["kernel_ms"][k] for k in ("forward_chain", "backward_chain")}, "status", d["status"])
PY
done; done
