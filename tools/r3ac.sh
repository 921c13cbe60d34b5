mkdir -p gpurun_out/r3ac
for N in 2 4 8; do for V in init trained; do python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 40 --emulate-world $N --variant $V > gpurun_out/r3ac/emu_${N}_$V.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3ac/emu_${N}_$V.json')); print('emulate-world $N $V:', d['ms_per_step'], d['kernel_ms'])"; done; done
for V in init trained; do EGR_RAYS_PER_TASK=64 python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 40 --emulate-world 8 --variant $V > gpurun_out/r3ac/emu_8_${V}_r64.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3ac/emu_8_${V}_r64.json')); print('emulate-world 8 $V 8x8 tasks:', d['ms_per_step'], d['kernel_ms'])"; done
