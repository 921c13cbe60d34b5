#!/bin/bash
# Register / LDS / spill report of the hot kernels (compiler view; works without a GPU).
cd "$(dirname "$0")/../editable-gaussian-reflections_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c trace.hip -o /tmp/egr_trace_res.o 2>&1 |
    grep -A11 "Function Name: .*k_\(forward\|backward\)" | sed 's/.*remark: //; s/ \[-Rpass.*//' |
    grep "Function Name\|VGPRs:\|Spill\|Occupancy\|LDS Size\|SGPRs:\|ScratchSize"
