mkdir -p gpurun_out/r3l
SWEEP_ARGS=" " tools/sweep.sh r3l "EGR_TABLE_EVICT=0" "EGR_TABLE_EVICT=1" 2>&1 | tee gpurun_out/r3l/sweep.txt
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_multirank.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3l/tests.log 2>&1; tail -4 gpurun_out/r3l/tests.log
