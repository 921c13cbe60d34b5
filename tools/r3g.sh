mkdir -p gpurun_out/r3g tools/ab_new
C=editable-gaussian-reflections_amd/csrc
for f in trace.hip forward_task.inc forward_decl.inc; do cp $C/$f tools/ab_new/$f; done
run() { # name
  touch $C/trace.hip
  python -c "import importlib; importlib.import_module('editable-gaussian-reflections_amd.build').build_all()" > gpurun_out/r3g/build_$1.log 2>&1 || { echo "$1 BUILD FAILED"; tail -5 gpurun_out/r3g/build_$1.log; return; }
  python bench.py --no-cpu-baseline --steps 60 --warmup 40 > gpurun_out/r3g/bench_$1.json 2> gpurun_out/r3g/bench_$1.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3g/bench_$1.json")); o=d.get("other_variant") or {}
    print("$1:", d["value"], d["ms_per_step"], {k:d["kernel_ms"][k] for k in ("forward_chain","backward_chain")}, "status", d["status"], "| other", o.get("value"), (o.get("kernel_ms") or {}).get("forward_chain"), (o.get("kernel_ms") or {}).get("backward_chain"))
except Exception as e:
    print("$1: FAILED", e)
PY
}
for f in trace.hip forward_task.inc forward_decl.inc; do cp tools/ab_prev/$f $C/$f; done
run prev
for f in trace.hip forward_task.inc forward_decl.inc; do cp tools/ab_new/$f $C/$f; done
run new
python -m pytest tests/test_hip_parity.py tests/test_hip_configs.py tests/test_hip_sequences.py -m gpu -q -s -x -p no:cacheprovider > gpurun_out/r3g/tests.log 2>&1; tail -4 gpurun_out/r3g/tests.log
grep -a -o "REPORT.*" gpurun_out/r3g/tests.log > gpurun_out/r3g/reports.txt
