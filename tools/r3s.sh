mkdir -p gpurun_out/r3s
for ST in 1 2; do
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TASK_TIMES=$ST python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3s/build_$ST.log 2>&1 || tail -5 gpurun_out/r3s/build_$ST.log
for C in 10 11; do echo "== step $ST init call $C"; VARIANT=init TT_STEP=$ST CALL=$C python tools/task_times.py 2>&1 | grep -a "heaviest\|tasks "; done
done
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3s/build_final.log 2>&1
