#!/bin/bash
# Usage (GPU box): tools/rh_emu.sh <tag> "<build env>" ... - per build setting: rank 0 of an emulated 8-way partition, remote help off / on / off / on, both clouds (forward / backward chain ms)
T=$1; shift
mkdir -p gpurun_out/$T
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$T/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$T/build_$NAME.log; continue; }
  for rh in 0 1 0 1; do for V in init trained; do
    EGR_PRINT_RH_STATS=1 EGR_REMOTE_HELP=$rh timeout 300 python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world ${RH_WORLD:-8} --variant $V > gpurun_out/$T/b.json 2> gpurun_out/$T/b.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$T/b.json"))
    print("$CFG rh=$rh $V: iteration", d["ms_per_step"], "fwd", d["kernel_ms"]["forward_chain"], "bwd", d["kernel_ms"]["backward_chain"], "status", d["status"])
except Exception as e:
    print("$CFG rh=$rh $V FAILED", e); print(open("gpurun_out/$T/b.err").read()[-800:])
PY
  done; done
  grep "remote help" gpurun_out/$T/b.err | grep -v "offers 0" | tail -2 | cut -c1-420
done
