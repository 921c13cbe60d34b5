#!/bin/bash
# Usage (GPU box): tools/chain_lpt.sh - per-task stamps of the whole forward chain (EGR_TASK_TIMES=9 build): span, list schedules of the same durations
# in other orders (longest first; descending primary leaf count), for both clouds
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TASK_TIMES=9 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
for V in ${VARIANTS:-init trained}; do echo "== $V"; CHAIN_DUMP=gpurun_out/chain_dump VARIANT=$V CALLS=10 EGR_TASK_TIMES=9 python tools/chain_times.py 2>&1 | grep -a "call\|span\|leaves\|tiles with"; done
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
