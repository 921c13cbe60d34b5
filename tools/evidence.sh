#!/bin/bash
# Usage (GPU box): tools/evidence.sh <tag> - the round's evidence set into gpurun_out/<tag>: GPU suite with the measured levels, smoke, the default bench line and
# config B, rocprofv3 kernel stats + PMC passes (tools/profile.sh), kernel resources, emulated partitions (rank 0 of 2 / 4 / 8; all ranks of 8), atomic rate, soak
T=${1:-evidence}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/$T/tests.log 2>&1; tail -3 gpurun_out/$T/tests.log
grep -a -o "REPORT.*" gpurun_out/$T/tests.log > gpurun_out/$T/parity_levels.txt
grep -a "passed\|failed" gpurun_out/$T/tests.log | tail -1 > gpurun_out/$T/gpu_suite_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/$T/gpu_suite_summary.txt
# the profiler passes FIRST: the bench lines below quote this run's HBM traffic (bench.py reads profiles/<round>/pmc_summary.json - refreshed here, in the box's
# working copy, before the lines are printed; round 5's line carried the previous run's counters)
bash tools/profile.sh $T/prof > gpurun_out/$T/profile.log 2>&1
for f in rocprofv3_kernel_stats_init.csv rocprofv3_kernel_stats_trained.csv bench_line_under_rocprof_init.json bench_line_under_rocprof_trained.json pmc_summary.json; do cp gpurun_out/$T/prof/$f gpurun_out/$T/ 2>/dev/null; done
ROUND_DIR=profiles/$(python -c "import bench; print(bench.ROUND)")
mkdir -p $ROUND_DIR && [ -s gpurun_out/$T/pmc_summary.json ] && cp gpurun_out/$T/pmc_summary.json $ROUND_DIR/pmc_summary.json
python bench.py > gpurun_out/$T/bench_line_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --config B > gpurun_out/$T/bench_line_config_B.json 2> gpurun_out/$T/bench_B.err
bash tools/kernel_resources.sh > gpurun_out/$T/kernel_resources.txt 2>&1
bash tools/emu_all.sh $T > /dev/null 2>&1
bash tools/emu_ranks.sh $T 8 > /dev/null 2>&1
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null && /tmp/atomic_rate > gpurun_out/$T/atomic_rate.txt 2>&1
python tools/soak.py > gpurun_out/$T/soak.txt 2>&1; tail -2 gpurun_out/$T/soak.txt
rm -rf gpurun_out/$T/prof/trace_* gpurun_out/$T/prof/pmc_*_[0-9] 2>/dev/null
ls gpurun_out/$T | head -40
