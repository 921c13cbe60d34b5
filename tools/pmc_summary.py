"""Folds the rocprofv3 --pmc passes of tools/profile.sh into {workload: {kernel: {counter: sum over the dispatch}}} (raw counter
units; FETCH_SIZE / WRITE_SIZE are KiB). Workload = <config>_<variant> (C_init, C_trained, B_init), one directory pmc_<workload>_<i>
per pass. Dispatches of the LAST launch: the fused per-tile chains k_forward_chain / k_backward_chain, k_grad_gather, k_finish.
Usage: python tools/pmc_summary.py <dir with pmc_*/...counter_collection.csv>"""
import collections, csv, glob, json, os, re, sys

out = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in sorted(glob.glob(sys.argv[1] + "/pmc_*")):
    mo = re.match(r"pmc_([BC]_[a-z]+)_\d+$", os.path.basename(d))
    if not mo or not os.path.isdir(d):
        continue
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        items = sorted(per.items())
        pick = lambda pat: [v for (dd, nm), v in items if pat in nm][-1:]
        for name, pat in (("forward_chain", "k_forward_chain"), ("backward_chain", "k_backward_chain"), ("backward_grad_gather", "k_grad_gather"), ("write_outputs", "k_finish")):
            for v in pick(pat):
                out[mo.group(1)][name].update(v)
print(json.dumps(out, indent=1))
