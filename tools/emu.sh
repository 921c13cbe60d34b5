#!/bin/bash
# Usage (GPU box): tools/emu.sh <tag> [world...]  - rank 0 of an emulated N-way tile partition on this one GPU (no collective), both clouds
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for N in "${@:-8}"; do for V in init trained; do
  python bench.py --no-cpu-baseline --no-second-variant --steps 60 --warmup 40 --primary-steps 0 --emulate-world $N --variant $V ${EMU_ARGS} 2>/dev/null | tail -1 > gpurun_out/$TAG/emu.json
  python - <<PY | tee -a gpurun_out/$TAG/emulated_partition_rank0.jsonl
import json
d = json.load(open("gpurun_out/$TAG/emu.json"))
print(json.dumps({"world": $N, "variant": "$V", "ms_per_iteration": d["ms_per_step"], "kernel_ms": d["kernel_ms"], "env": "${EMU_ENV}"}))
PY
done; done
