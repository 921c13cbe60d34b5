#!/bin/bash
# Usage (GPU box): tools/sweep_emu4.sh <tag> "<worlds>" "BUILDENV1" "BUILDENV2" ... - rank 0 of emulated partitions (team help on) per build-time setting, both
# clouds, two runs each; "-" = the library as it is, "run:VAR=x" = an environment setting of the run (neither rebuilds)
TAG=$1; WORLDS=$2; shift 2
mkdir -p gpurun_out/$TAG
for B in "$@"; do
  E=""
  if [ "${B#run:}" != "$B" ]; then E="${B#run:}"  # "run:VAR=x": an environment setting of the RUN, no rebuild
  elif [ "$B" != "-" ]; then
    touch editable-gaussian-reflections_amd/csrc/trace.hip
    env $B python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build.log 2>&1 || { echo "$B: BUILD FAILED"; tail -5 gpurun_out/$TAG/build.log; continue; }
  fi
  for W in $WORLDS; do for V in init trained; do for rep in 1 2; do
    env $E python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $W --variant $V 2>/dev/null | tail -1 > gpurun_out/$TAG/emu.json
    python - <<PY | tee -a gpurun_out/$TAG/results.txt
import json
d = json.load(open("gpurun_out/$TAG/emu.json"))
print("$B | world $W $V:", d["ms_per_step"], {k: d["kernel_ms"][k] for k in ("forward_chain", "backward_chain")}, "status", d["status"])
PY
  done; done; done
done
