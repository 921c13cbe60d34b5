mkdir -p gpurun_out/r3m
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_multirank.py tests/test_hip_sequences.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r3m/tests.log 2>&1; tail -5 gpurun_out/r3m/tests.log
for R in 64 32 16; do for V in init trained; do EGR_RAYS_PER_TASK=$R python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 40 --emulate-world 8 --variant $V > gpurun_out/r3m/emu8_${R}_$V.json 2>gpurun_out/r3m/emu8_${R}_$V.err; python -c "
import json; d=json.load(open('gpurun_out/r3m/emu8_${R}_$V.json')); print('emulate-world 8 rays/task $R $V:', d['ms_per_step'], d['kernel_ms'], 'status', d['status'])"; done; done
for R in 32 16; do EGR_RAYS_PER_TASK=$R python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 40 --emulate-world 2 --variant trained > gpurun_out/r3m/emu2_${R}.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3m/emu2_${R}.json')); print('emulate-world 2 rays/task $R trained:', d['ms_per_step'], d['kernel_ms'])"; done
EGR_RAYS_PER_TASK=16 python bench.py --no-cpu-baseline --steps 40 --warmup 40 > gpurun_out/r3m/full_16.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3m/full_16.json')); print('whole image, 16 rays/task:', d['value'], d['value_trained_like'])"
