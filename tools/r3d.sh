mkdir -p gpurun_out/r3d
SWEEP_ARGS=" " tools/sweep.sh r3d "EGR_PIPELINE=0" "EGR_PIPELINE=1" "EGR_PIPELINE=1 EGR_GPOP=6" "EGR_PIPELINE=0 EGR_GPOP=6" "EGR_PIPELINE=1 EGR_GPOP=8" "EGR_PIPELINE=1 EGR_PAIR_PRIMARY=1" "EGR_PIPELINE=0 EGR_PAIR_PRIMARY=1" 2>&1 | tee gpurun_out/r3d/sweep.txt
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3d/build_final.log 2>&1
python -m pytest tests/test_hip_parity.py tests/test_hip_sequences.py -m gpu -q -s -x -p no:cacheprovider > gpurun_out/r3d/tests.log 2>&1; tail -4 gpurun_out/r3d/tests.log
