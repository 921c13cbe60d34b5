mkdir -p gpurun_out/r3r
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -p no:cacheprovider > gpurun_out/r3r/tests.log 2>&1; tail -3 gpurun_out/r3r/tests.log
python tools/launch_times.py 300 init 2>&1 | tail -1
python tools/launch_times.py 300 trained 2>&1 | tail -1
python bench.py --no-cpu-baseline --steps 100 --warmup 50 > gpurun_out/r3r/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3r/bench.json')); print('bench', d['value'], d['value_trained_like'], d['ms_per_step'], d['kernel_ms'], d['other_variant']['kernel_ms'])"
