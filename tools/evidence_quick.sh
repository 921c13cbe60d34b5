#!/bin/bash
# Usage (GPU box): tools/evidence_quick.sh <tag>  - the short evidence run: GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the
# headline variant, rank 0 of an emulated 8-way partition. (tools/evidence.sh adds the PMC passes, config B, N = 2 / 4 and the soak.)
T=${1:-r4quick}
mkdir -p gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q -s -p no:cacheprovider ) > gpurun_out/$T/tests.log 2>&1; tail -4 gpurun_out/$T/tests.log
grep -a -o "REPORT.*" gpurun_out/$T/tests.log > gpurun_out/$T/parity_levels.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/$T/bench_line_default.json 2> gpurun_out/$T/bench_default.err; cut -c1-400 gpurun_out/$T/bench_line_default.json
mkdir -p gpurun_out/$T/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof/trace_init -o t -- python bench.py --strands 1 --steps 100 --warmup 300 --prewarm-seconds 3 --no-cpu-baseline --no-second-variant --primary-steps 0 --variant init > gpurun_out/$T/prof/bench_under_rocprof_init.log 2>&1
grep -a "^{" gpurun_out/$T/prof/bench_under_rocprof_init.log | tail -1 > gpurun_out/$T/prof/bench_line_under_rocprof_init.json
find gpurun_out/$T/prof/trace_init -name "*kernel_stats.csv" -exec cp {} gpurun_out/$T/rocprofv3_kernel_stats_init.csv \;
rm -rf gpurun_out/$T/prof/trace_init
head -8 gpurun_out/$T/rocprofv3_kernel_stats_init.csv
bash tools/emu.sh $T 8
