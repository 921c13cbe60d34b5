mkdir -p gpurun_out/r3c
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -s -x -p no:cacheprovider > gpurun_out/r3c/tests.log 2>&1; tail -8 gpurun_out/r3c/tests.log
grep -a -o "REPORT.*" gpurun_out/r3c/tests.log > gpurun_out/r3c/reports.txt
SWEEP_ARGS=" " tools/sweep.sh r3c "EGR_PAIR_WALK=0" "EGR_PAIR_WALK=1" "EGR_PAIR_WALK=1 EGR_GPOP=2" "EGR_PAIR_WALK=1 EGR_GPOP=3" 2>&1 | tee gpurun_out/r3c/sweep.txt
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TRAVERSAL_STATS=1 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3c/build_stats.log 2>&1
EGR_PRINT_TRAVERSAL_STATS=1 python tools/stats_run.py > gpurun_out/r3c/stats.txt 2>&1
tail -12 gpurun_out/r3c/stats.txt
