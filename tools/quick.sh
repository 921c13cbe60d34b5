#!/bin/bash
# Usage (GPU box): tools/quick.sh <tag> [world...] - GPU suite (stops at the first failure), the default bench line without the CPU leg, rank 0 of emulated partitions
T=${1:-q}; shift
mkdir -p gpurun_out/$T
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$T/tests.log 2>&1; tail -4 gpurun_out/$T/tests.log | head -2
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$T/bench.json")); o=d.get("other_variant") or {}
    print("init", d["value"], d["ms_per_step"], d["kernel_ms"], "status", d["status"], "primary-only", d.get("value_primary_only"))
    print("trained", o.get("value"), o.get("ms_per_step"), o.get("kernel_ms"), "status", o.get("status"))
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/$T/bench.err").read()[-2000:])
PY
[ $# -gt 0 ] && bash tools/emu.sh $T "$@"
