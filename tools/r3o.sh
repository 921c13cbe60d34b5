mkdir -p gpurun_out/r3o
SWEEP_ARGS=" " tools/sweep.sh r3o "EGR_PACKET_PREFETCH=0" "EGR_PACKET_PREFETCH=1" 2>&1 | tee gpurun_out/r3o/sweep.txt
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py -m gpu -q -p no:cacheprovider > gpurun_out/r3o/tests.log 2>&1; tail -3 gpurun_out/r3o/tests.log
python bench.py --config B --no-cpu-baseline --steps 60 --warmup 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config B', d['value'], d['kernel_ms'])"
