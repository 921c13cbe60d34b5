#!/bin/bash
# Usage (GPU box): tools/task_split.sh <step> [variant] - per-task stamps of ONE forward step (EGR_TASK_TIMES=<step> build): walk vs selection + compositing of the heaviest tasks
STEP=${1:-1}; V=${2:-init}
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TASK_TIMES=$STEP python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
VARIANT=$V TT_STEP=$STEP CALL=10 python tools/task_times.py 2>&1 | grep -a "heaviest\|longest list\|walk share\|tasks "
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
