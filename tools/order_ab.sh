#!/bin/bash
# Usage (GPU box): tools/order_ab.sh - whole-image bench with the forward order off / on / automatic (EGR_FORWARD_ORDER), interleaved
for M in 0 1 -1 0 1 -1; do EGR_FORWARD_ORDER=$M python bench.py --no-cpu-baseline --steps 60 --warmup 40 --primary-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_variant']; print('forward order $M', d['value'], d['ms_per_step'], d['kernel_ms'], '|', o['value'], o['ms_per_step'], {k: o['kernel_ms'][k] for k in o['kernel_ms'] if 'chain' in k or 'prewalk' in k})"; done
