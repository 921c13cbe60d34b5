mkdir -p gpurun_out/r3b
python -m pytest tests/test_hip_configs.py tests/test_hip_sequences.py tests/test_c_abi_direct.py tests/test_hip_parity.py tests/test_fused_step.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3b/tests.log 2>&1; tail -8 gpurun_out/r3b/tests.log
grep -a -o "REPORT.*" gpurun_out/r3b/tests.log > gpurun_out/r3b/reports.txt
SWEEP_ARGS=" " tools/sweep.sh r3b "EGR_SAH_COLLAPSE=0" "EGR_SAH_COLLAPSE=1" 2>&1 | tee gpurun_out/r3b/sweep.txt
for S in 0 1; do
  touch editable-gaussian-reflections_amd/csrc/trace.hip editable-gaussian-reflections_amd/csrc/bvh.hip
  EGR_SAH_COLLAPSE=$S EGR_TRAVERSAL_STATS=1 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3b/build_stats_$S.log 2>&1
  EGR_PRINT_TRAVERSAL_STATS=1 python tools/stats_run.py > gpurun_out/r3b/stats_$S.txt 2>&1
  tail -12 gpurun_out/r3b/stats_$S.txt
done
