#!/bin/bash
# Usage (GPU box): tools/sweep_help.sh <tag> "ENV=a" ... - like tools/sweep.sh, with team help ON for the whole image (bench.py --team-help 1)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for CFG in "$@"; do
  NAME=$(echo "$CFG" | tr ' =' '__')
  touch editable-gaussian-reflections_amd/csrc/trace.hip
  env $CFG python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$TAG/build_$NAME.log 2>&1 || { echo "$CFG: BUILD FAILED"; continue; }
  env $CFG python bench.py --no-cpu-baseline --steps 60 --warmup 40 --primary-steps 0 --team-help 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_variant']; print('$CFG (help on):', d['value'], d['ms_per_step'], d['kernel_ms']['forward_chain'], d['kernel_ms']['backward_chain'], 'status', d['status'], '|', o['value'], o['ms_per_step'], o['kernel_ms']['forward_chain'])"
done
