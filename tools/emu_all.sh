#!/bin/bash
# Usage (GPU box): tools/emu_all.sh <tag> - rank 0 of emulated 2 / 4 / 8-way partitions on this one GPU (no collective), both clouds, with team help (what
# bench.py does for ranks of a partition) and, for 8, without
T=${1:-emu}; mkdir -p gpurun_out/$T; : > gpurun_out/$T/emulated_partition_rank0.jsonl
one() { # world variant team_help
  python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $1 --variant $2 --team-help $3 > gpurun_out/$T/emu.json 2>/dev/null
  python - <<PY >> gpurun_out/$T/emulated_partition_rank0.jsonl
import json
d = json.load(open("gpurun_out/$T/emu.json"))
print(json.dumps({"world": $1, "variant": "$2", "team_help": bool($3), "ms_per_iteration": d["ms_per_step"], "forward_chain_ms": d["kernel_ms"]["forward_chain"], "backward_chain_ms": d["kernel_ms"]["backward_chain"], "kernel_ms": d["kernel_ms"]}))
PY
}
python bench.py --no-cpu-baseline --steps 60 --warmup 40 --primary-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'world': 1, 'variant': 'init', 'ms_per_iteration': d['ms_per_step'], 'kernel_ms': d['kernel_ms']})); o=d['other_variant']; print(json.dumps({'world': 1, 'variant': 'trained', 'ms_per_iteration': o['ms_per_step'], 'kernel_ms': o['kernel_ms']}))" >> gpurun_out/$T/emulated_partition_rank0.jsonl
for N in 2 4 8; do for V in init trained; do one $N $V 1; done; done
for V in init trained; do one 8 $V 0; done
cat gpurun_out/$T/emulated_partition_rank0.jsonl
