touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TASK_TIMES=9 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
CALLS=9,10,11,30 python tools/chain_times.py 2>&1 | grep "^call"
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > /dev/null 2>&1
