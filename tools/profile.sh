#!/bin/bash
# Usage (on the GPU box): tools/profile.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of `bench.py --strands 1` (kernels run one at a time, so the per-kernel average durations
#    are exclusive and agree with the bench line's roofline.avg_kernel_ms); 2) separate, time-boxed --pmc passes on ONE launch of
#    each bench workload (EGR_STRANDS=1: one dispatch per chain): FETCH_SIZE and WRITE_SIZE for C_init, C_trained and B_init, the
#    cache / TA / SQ groups for the headline workload. PMC is never combined with trace domains other than --kernel-trace.
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
# one process per opacity variant, so that a kernel's average in the stats file belongs to ONE workload (the headline line first);
# 400+ launches, so that the first few cold ones (clock ramp: up to 3x the steady duration) do not carry the average
for VAR in init trained; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$VAR -o t -- python bench.py --strands 1 --steps 100 --warmup 300 --prewarm-seconds 3 --no-cpu-baseline --no-second-variant --primary-steps 0 --variant $VAR > $OUT/bench_under_rocprof_$VAR.log 2>&1
  grep -a "^{" $OUT/bench_under_rocprof_$VAR.log | tail -1 > $OUT/bench_line_under_rocprof_$VAR.json
  find $OUT/trace_$VAR -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_$VAR.csv \;
done
pass() { # workload-config workload-variant index counters...
  local CFG=$1 VAR=$2 IDX=$3; shift 3
  ( time PMC_CONFIG=$CFG PMC_VARIANT=$VAR EGR_STRANDS=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${CFG}_${VAR}_$IDX -o p -- python tools/pmc_run.py ) > $OUT/pmc_${CFG}_${VAR}_$IDX.log 2>&1
}
for WL in "C init" "C trained" "B init"; do
  set -- $WL
  pass $1 $2 1 FETCH_SIZE
  pass $1 $2 2 WRITE_SIZE
done
pass C init 3 TCC_HIT TCC_MISS TCC_REQ
pass C init 4 TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES
pass C init 5 TA_TA_BUSY TCP_GATE_EN1 TCP_PENDING_STALL_CYCLES
pass C init 6 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass C init 7 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass C init 8 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU
pass C trained 3 TCC_HIT TCC_MISS TCC_REQ
pass C trained 4 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass C trained 5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass C trained 6 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
ls -la $OUT | head -40
