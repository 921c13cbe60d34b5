mkdir -p gpurun_out/r3h
SWEEP_ARGS=" " tools/sweep.sh r3h "EGR_COMPOSITE_PREFETCH=0" "EGR_COMPOSITE_PREFETCH=1" 2>&1 | tee gpurun_out/r3h/sweep.txt
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3h/tests.log 2>&1; tail -6 gpurun_out/r3h/tests.log
grep -a -o "REPORT.*" gpurun_out/r3h/tests.log > gpurun_out/r3h/reports.txt
