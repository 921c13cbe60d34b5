mkdir -p gpurun_out/r3k
SWEEP_ARGS=" " tools/sweep.sh r3k "EGR_BWD_WAVES=3" "EGR_BWD_WAVES=4" "EGR_GT_SLOTS=128" "EGR_GT_SLOTS=128 EGR_BWD_WAVES=2" "EGR_BWD_WAVES=2" 2>&1 | tee gpurun_out/r3k/sweep.txt
touch editable-gaussian-reflections_amd/csrc/trace.hip
python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3k/build_final.log 2>&1
for N in 2 4 8; do for V in init trained; do python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 40 --emulate-world $N --variant $V > gpurun_out/r3k/emu_${N}_$V.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3k/emu_${N}_$V.json')); print('emulate-world $N $V:', d['ms_per_step'], d['kernel_ms'])"; done; done
