"""Folds the rocprofv3 --pmc passes of tools/profile.sh into {kernel step name: {counter: sum over the dispatch}} (raw counter
units; FETCH_SIZE / WRITE_SIZE are KiB). Dispatch order inside one iteration: k_forward x3 (steps 0,1,2), k_backward x3
(steps 2,1,0), k_bucket_reduce, k_grad_gather. Usage: python tools/pmc_summary.py <dir with pmc*/...counter_collection.csv>"""
import collections, csv, glob, json, sys

out = collections.defaultdict(dict)
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    fwd = [v for (d, n), v in sorted(per.items()) if "k_forward" in n][-3:]
    bwd = [v for (d, n), v in sorted(per.items()) if "k_backward" in n][-3:]
    red = [v for (d, n), v in sorted(per.items()) if "k_bucket_reduce" in n or "k_log_apply" in n][-1:]
    names = [f"forward_step{i}" for i in range(len(fwd))] + [f"backward_step{2 - i}" for i in range(len(bwd))] + ["backward_bucket_reduce"] * len(red)
    for name, v in zip(names, fwd + bwd + red):
        out[name].update(v)
print(json.dumps(out, indent=1))
