#!/bin/bash
# Usage (GPU box): tools/emu_ranks.sh <tag> [world] - EVERY rank of an emulated N-way tile partition (default 8), one after the other on this one GPU (no
# collective), both clouds, with and without team help; the iteration of an N-GPU run lasts as long as its slowest rank: the summary prints the MAX over ranks
T=${1:-emu_ranks}; N=${2:-8}; mkdir -p gpurun_out/$T; OUT=gpurun_out/$T/emulated_partition_all_ranks.jsonl; : > $OUT
for V in init trained; do for H in 0 1; do for R in $(seq 0 $((N-1))); do
  python bench.py --no-cpu-baseline --no-second-variant --steps 40 --warmup 30 --primary-steps 0 --emulate-world $N --emulate-rank $R --variant $V --team-help $H 2>/dev/null | tail -1 > gpurun_out/$T/emu.json
  python - <<PY >> $OUT
import json
d = json.load(open("gpurun_out/$T/emu.json"))
print(json.dumps({"world": $N, "rank": $R, "variant": "$V", "team_help": bool($H), "ms_per_iteration": d["ms_per_step"], "forward_chain_ms": d["kernel_ms"]["forward_chain"], "backward_chain_ms": d["kernel_ms"]["backward_chain"], "multi_gpu": d.get("multi_gpu"), "kernel_ms": d["kernel_ms"]}))
PY
done; done; done
python - <<PY | tee gpurun_out/$T/emulated_partition_all_ranks_summary.txt
import json, collections
rows = [json.loads(l) for l in open("$OUT")]
g = collections.defaultdict(list)
for r in rows: g[(r["variant"], r["team_help"])].append(r)
for k, v in sorted(g.items()):
    it = [r["ms_per_iteration"] for r in v]; f = [r["forward_chain_ms"] for r in v]; b = [r["backward_chain_ms"] for r in v]
    rep = [r["multi_gpu"]["replicated_ms"] for r in v if r.get("multi_gpu")]
    print(f"world $N {k[0]:8s} team_help={k[1]!s:5s} iteration ms: max over ranks {max(it):.3f} (rank {it.index(max(it))}), min {min(it):.3f}, mean {sum(it)/len(it):.3f} | forward chain max {max(f):.3f} | backward chain max {max(b):.3f} | replicated (non-chain) max {max(rep) if rep else float('nan'):.3f}")
PY
