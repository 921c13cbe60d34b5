mkdir -p gpurun_out/r3j
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r3j/tests.log 2>&1; tail -3 gpurun_out/r3j/tests.log
grep -a -o "REPORT.*" gpurun_out/r3j/tests.log > gpurun_out/r3j/parity_levels.txt
python bench.py > gpurun_out/r3j/bench_line_default.json 2> gpurun_out/r3j/bench_default.err
python bench.py --config B > gpurun_out/r3j/bench_line_config_B.json 2> gpurun_out/r3j/bench_B.err
bash tools/profile.sh r3j/prof
bash tools/kernel_resources.sh > gpurun_out/r3j/kernel_resources.txt 2>&1
