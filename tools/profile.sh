#!/bin/bash
# Usage (on the GPU box): tools/profile.sh <tag> [bench args...]
# Collects: kernel-trace stats + separate PMC passes (never combined with other trace domains) into gpurun_out/<tag>/
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/trace.log 2>&1
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
           "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_READ" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- python bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.csv" | head -40
