T=${1:-r4final}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/$T/tests.log 2>&1; tail -3 gpurun_out/$T/tests.log
grep -a -o "REPORT.*" gpurun_out/$T/tests.log > gpurun_out/$T/parity_levels.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/$T/bench_line_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --config B > gpurun_out/$T/bench_line_config_B.json 2> gpurun_out/$T/bench_B.err
bash tools/profile.sh $T/prof > gpurun_out/$T/profile.log 2>&1
bash tools/kernel_resources.sh > gpurun_out/$T/kernel_resources.txt 2>&1
python bench.py > gpurun_out/$T/bench_line_default_with_traffic.json 2> /dev/null
ls gpurun_out/$T/prof | head -50
# rank 0 of an emulated N-way tile partition on this one GPU (no collective): ms per iteration and the two chains
: > gpurun_out/$T/emulated_partition_rank0.jsonl
for N in 2 4 8; do for V in init trained; do
  python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $N --variant $V > gpurun_out/$T/emu.json 2>/dev/null
  python - <<PY >> gpurun_out/$T/emulated_partition_rank0.jsonl
import json
d = json.load(open("gpurun_out/$T/emu.json"))
print(json.dumps({"world": $N, "variant": "$V", "team_help": True, "ms_per_iteration": d["ms_per_step"], "forward_chain_ms": d["kernel_ms"]["forward_chain"], "backward_chain_ms": d["kernel_ms"]["backward_chain"], "kernel_ms": d["kernel_ms"]}))
PY
done; done
# the same rank without team help (bench.py turns it on for ranks of a partition)
for V in init trained; do
  python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world 8 --variant $V --team-help 0 > gpurun_out/$T/emu.json 2>/dev/null
  python - <<PY >> gpurun_out/$T/emulated_partition_rank0.jsonl
import json
d = json.load(open("gpurun_out/$T/emu.json"))
print(json.dumps({"world": 8, "variant": "$V", "team_help": False, "ms_per_iteration": d["ms_per_step"], "forward_chain_ms": d["kernel_ms"]["forward_chain"], "backward_chain_ms": d["kernel_ms"]["backward_chain"], "kernel_ms": d["kernel_ms"]}))
PY
done
cat gpurun_out/$T/emulated_partition_rank0.jsonl
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null && /tmp/atomic_rate > gpurun_out/$T/atomic_rate.txt 2>&1
python tools/soak.py > gpurun_out/$T/soak.txt 2>&1; tail -2 gpurun_out/$T/soak.txt
