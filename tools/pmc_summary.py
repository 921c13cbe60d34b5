"""Folds the rocprofv3 --pmc passes of tools/profile.sh into {kernel: {counter: sum over the dispatch}} (raw counter units;
FETCH_SIZE / WRITE_SIZE are KiB). Dispatches of the LAST iteration: the fused per-tile chains k_forward_chain / k_backward_chain and
k_grad_gather. Usage: python tools/pmc_summary.py <dir with pmc*/...counter_collection.csv>"""
import collections, csv, glob, json, sys

out = collections.defaultdict(dict)
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    items = sorted(per.items())
    pick = lambda pat: [v for (d, nm), v in items if pat in nm][-1:]
    for name, pat in (("forward_chain", "k_forward_chain"), ("backward_chain", "k_backward_chain"), ("backward_grad_gather", "k_grad_gather")):
        for v in pick(pat):
            out[name].update(v)
print(json.dumps(out, indent=1))
