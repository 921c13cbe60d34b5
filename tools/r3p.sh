mkdir -p gpurun_out/r3p
for ST in 0 1; do
touch editable-gaussian-reflections_amd/csrc/trace.hip
EGR_TASK_TIMES=$ST python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3p/build_$ST.log 2>&1 || tail -5 gpurun_out/r3p/build_$ST.log
for V in init trained; do for Wd in 1 8; do echo "== step $ST $V world $Wd"; VARIANT=$V TT_STEP=$ST EMU_WORLD=$Wd EGR_RAYS_PER_TASK=64 python tools/task_times.py 2>&1 | grep -v amdgpu.ids; done; done
done
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3p/build_final.log 2>&1
