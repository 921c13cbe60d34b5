cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3q; mkdir -p $OUT
for VAR in init trained; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$VAR -o t -- python bench.py --strands 1 --steps 100 --warmup 300 --prewarm-seconds 3 --no-cpu-baseline --no-second-variant --variant $VAR > $OUT/bench_under_rocprof_$VAR.log 2>&1
  grep -a "^{" $OUT/bench_under_rocprof_$VAR.log | tail -1 > $OUT/bench_line_under_rocprof_$VAR.json
  find $OUT/trace_$VAR -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_$VAR.csv \;
  head -3 $OUT/rocprofv3_kernel_stats_$VAR.csv | cut -c1-150
  python -c "
import json; d=json.load(open('$OUT/bench_line_under_rocprof_$VAR.json')); print('$VAR line:', d['value'], d['kernel_ms'])"
done
