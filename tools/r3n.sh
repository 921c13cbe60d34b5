mkdir -p gpurun_out/r3n
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_PSTK=64 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3n/build_pstk64.log 2>&1 || tail -5 gpurun_out/r3n/build_pstk64.log
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py -m gpu -q -p no:cacheprovider > gpurun_out/r3n/tests_pstk64.log 2>&1; echo "PSTK=64 (pair-stack spill path in use):"; tail -3 gpurun_out/r3n/tests_pstk64.log
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3n/build_final.log 2>&1
python -m pytest tests/test_hip_parity.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -a -o "REPORT default_statistic.*\|[0-9]* passed.*\|[0-9]* failed.*"
timeout 900 python tools/soak.py 600 > gpurun_out/r3n/soak.txt 2>&1; tail -6 gpurun_out/r3n/soak.txt
