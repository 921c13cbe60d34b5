mkdir -p gpurun_out/r3f
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TRAVERSAL_STATS=1 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3f/build_stats.log 2>&1 || tail -20 gpurun_out/r3f/build_stats.log
for V in init trained; do
  VARIANT=$V GRADS=1 EGR_PRINT_TRAVERSAL_STATS=1 python tools/stats_run.py > gpurun_out/r3f/stats_$V.txt 2>&1
  grep -a "egr stats\|rays" gpurun_out/r3f/stats_$V.txt
done
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3f/build_final.log 2>&1
python -m pytest tests/test_hip_sequences.py -m gpu -q -s -x -p no:cacheprovider -k config4 > gpurun_out/r3f/tests.log 2>&1; tail -4 gpurun_out/r3f/tests.log
