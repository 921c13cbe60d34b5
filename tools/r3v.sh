mkdir -p gpurun_out/r3z tools/ab_new
C=editable-gaussian-reflections_amd/csrc
cp $C/backward_task.inc tools/ab_new/backward_task.inc
run() {
  touch $C/trace.hip
  python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3z/build_$1.log 2>&1 || { echo "$1 BUILD FAILED"; tail -5 gpurun_out/r3z/build_$1.log; return; }
  python bench.py --no-cpu-baseline --steps 100 --warmup 50 > gpurun_out/r3z/bench_$1.json 2> gpurun_out/r3z/bench_$1.err
  python -c "
import json
d=json.load(open('gpurun_out/r3z/bench_$1.json')); o=d['other_variant']
print('$1:', d['value'], d['ms_per_step'], {k:d['kernel_ms'][k] for k in ('forward_chain','backward_chain')}, 'status', d['status'], '| other', o['value'], o['kernel_ms']['forward_chain'], o['kernel_ms']['backward_chain'])"
}
cp tools/ab_prev/backward_task.inc $C/backward_task.inc; run prev
cp tools/ab_new/backward_task.inc $C/backward_task.inc; run new
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -p no:cacheprovider > gpurun_out/r3z/tests.log 2>&1; tail -3 gpurun_out/r3z/tests.log
