mkdir -p gpurun_out/r3i
SWEEP_ARGS=" " tools/sweep.sh r3i "EGR_TWO_STAGE=0" "EGR_TWO_STAGE=1" 2>&1 | tee gpurun_out/r3i/sweep.txt
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TWO_STAGE=1 EGR_TRAVERSAL_STATS=1 python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3i/build_stats.log 2>&1
EGR_PRINT_TRAVERSAL_STATS=1 python tools/stats_run.py > gpurun_out/r3i/stats_two_stage.txt 2>&1; grep -a "egr stats" gpurun_out/r3i/stats_two_stage.txt | head -4
