T=${1:-r3final}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/$T/tests.log 2>&1; tail -3 gpurun_out/$T/tests.log
grep -a -o "REPORT.*" gpurun_out/$T/tests.log > gpurun_out/$T/parity_levels.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/$T/bench_line_default.json 2> gpurun_out/$T/bench_default.err
python bench.py --config B > gpurun_out/$T/bench_line_config_B.json 2> gpurun_out/$T/bench_B.err
bash tools/profile.sh $T/prof > gpurun_out/$T/profile.log 2>&1
bash tools/kernel_resources.sh > gpurun_out/$T/kernel_resources.txt 2>&1
python bench.py > gpurun_out/$T/bench_line_default_with_traffic.json 2> /dev/null
ls gpurun_out/$T/prof | head -50
