#!/bin/bash
# Usage (on the GPU box): tools/profile.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of `bench.py --strands 1` (kernels run one at a time, so the per-kernel average durations
#    are exclusive and agree with the bench line's roofline.avg_kernel_ms); 2) separate, time-boxed --pmc passes on ONE
#    iteration (EGR_STRANDS=1: one dispatch per step). PMC is never combined with other trace domains.
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --strands 1 --steps 20 --warmup 50 --no-cpu-baseline --no-second-variant > $OUT/bench_under_rocprof.log 2>&1
grep -a "^{" $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line_under_rocprof.json
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES" "TA_TA_BUSY TCP_GATE_EN1 TCP_PENDING_STALL_CYCLES"; do
  i=$((i+1))
  ( time EGR_STRANDS=1 timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- python tools/pmc_run.py ) > $OUT/pmc$i.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
cp $OUT/trace/*/t_kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats.csv \;
ls -la $OUT | head -20
