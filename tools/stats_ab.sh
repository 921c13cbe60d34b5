#!/bin/bash
# Usage (GPU box): tools/stats_ab.sh <tag> "ENV1=a" "ENV1=b" ... - rebuild with EGR_TRAVERSAL_STATS=1 and each build-time setting, print the per-phase
# cycle sums and visit counters of one full-size grad launch per cloud variant (VARIANTS, default "init trained")
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env EGR_TRAVERSAL_STATS=1 $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; tail -5 gpurun_out/$TAG/build_$NAME.log; continue; }
  for V in ${VARIANTS:-init trained}; do
    echo "== $CFG $V"
    env $CFG GRADS=1 VARIANT=$V EGR_PRINT_TRAVERSAL_STATS=1 python tools/stats_run.py 2>&1 | grep -a "egr stats\|rays" | tee gpurun_out/$TAG/stats_${NAME}_$V.txt
  done
done
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
