"""Per-launch durations of the two chain kernels from rocprofv3 --kernel-trace CSVs (tools/profile.sh) next to the bench line of the same process.
Usage: python tools/trace_summary.py <dir with trace_{init,trained}/ and bench_line_under_rocprof_{init,trained}.json>"""
import csv, glob, json, os, sys
import numpy as np
d = sys.argv[1]
print("# per-launch durations of the two chain kernels from the rocprofv3 kernel trace of the same processes as rocprofv3_kernel_stats_{init,trained}.csv")
print("# (python bench.py --strands 1 --steps 100 --warmup 300 --prewarm-seconds 3 --no-cpu-baseline --no-second-variant --variant V).")
print("# bench.py's roofline.avg_kernel_ms is the MEDIAN of its profile pass (HIP events on the launch stream).")
for var in ("init", "trained"):
    f = glob.glob(os.path.join(d, f"trace_{var}", "*kernel_trace.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    for name, key in (("k_forward_chain", "k_forward_chain<true, false>"), ("k_backward_chain", "k_backward_chain")):
        x = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if key in r["Kernel_Name"]])
        if x.size == 0:
            continue
        med = float(np.median(x))
        slow = np.nonzero(x > 1.2 * med)[0]
        print(f"{var:8s} {name:17s} launches {x.size}  median {med:.3f} ms  mean {x.mean():.3f} ms  min {x.min():.3f}  max {x.max():.3f}  launches > 1.2 x median: {slow.size} {np.round(x[slow], 2).tolist()[:12]}")
for var in ("init", "trained"):
    p = os.path.join(d, f"bench_line_under_rocprof_{var}.json")
    if os.path.exists(p):
        print(f"{var:8s} bench line of that process: kernel_ms {json.load(open(p))['kernel_ms']}")
