#!/bin/bash
# Usage (on the GPU box): tools/profile.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench command; 2) separate, time-boxed --pmc passes
#    (FETCH_SIZE / WRITE_SIZE / L2 hit) on ONE iteration. PMC is never combined with other trace domains.
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
tail -1 $OUT/bench_under_rocprof.log > $OUT/bench_line.json
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TOTAL_READ"; do
  i=$((i+1))
  ( time timeout 420 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- python tools/pmc_run.py ) > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.csv" | head -40
