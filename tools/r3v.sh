T=${1:-r3aa}
mkdir -p gpurun_out/$T tools/ab_new
C=editable-gaussian-reflections_amd/csrc
for f in forward_task.inc trace.hip; do cp $C/$f tools/ab_new/$f; done
run() {
  touch $C/trace.hip
  python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/$T/build_$1.log 2>&1 || { echo "$1 BUILD FAILED"; tail -5 gpurun_out/$T/build_$1.log; return; }
  python bench.py --no-cpu-baseline --steps 100 --warmup 50 > gpurun_out/$T/bench_$1.json 2> gpurun_out/$T/bench_$1.err
  python -c "
import json
d=json.load(open('gpurun_out/$T/bench_$1.json')); o=d['other_variant']
print('$1:', d['value'], d['ms_per_step'], {k:d['kernel_ms'][k] for k in ('forward_chain','backward_chain')}, 'status', d['status'], '| other', o['value'], o['kernel_ms']['forward_chain'], o['kernel_ms']['backward_chain'])"
}
for f in forward_task.inc trace.hip; do cp tools/ab_prev/$f $C/$f; done; run prev
for f in forward_task.inc trace.hip; do cp tools/ab_new/$f $C/$f; done; run new
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -p no:cacheprovider > gpurun_out/$T/tests.log 2>&1; tail -3 gpurun_out/$T/tests.log
