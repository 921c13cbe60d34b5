"""Folds the rocprofv3 --pmc passes of tools/profile.sh into {kernel step name: {counter: sum over the dispatch}} (raw counter
units; FETCH_SIZE / WRITE_SIZE are KiB). Dispatches of the LAST iteration: the fused per-tile chains k_forward_chain /
k_backward_chain (default), or with EGR_CHAIN=0 k_forward x3 (steps 0,1,2) and k_backward x3 (steps 2,1,0); then k_log_apply
(or k_bucket_reduce) and k_grad_gather. Usage: python tools/pmc_summary.py <dir with pmc*/...counter_collection.csv>"""
import collections, csv, glob, json, sys

out = collections.defaultdict(dict)
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        per.setdefault(key, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    items = sorted(per.items())
    pick = lambda pat, n: [v for (d, nm), v in items if pat in nm][-n:]
    named = []
    if pick("k_forward_chain", 1):
        named += list(zip(["forward_chain"], pick("k_forward_chain", 1))) + list(zip(["backward_chain"], pick("k_backward_chain", 1)))
    else:
        fwd, bwd = pick("k_forward<", 3) or pick("k_forward", 3), pick("k_backward<", 3) or pick("k_backward", 3)
        named += [(f"forward_step{i}", v) for i, v in enumerate(fwd)] + [(f"backward_step{2 - i}", v) for i, v in enumerate(bwd)]
    red = [v for (d, nm), v in items if "k_bucket_reduce" in nm or "k_log_apply" in nm][-1:]
    named += [("backward_bucket_reduce", v) for v in red]
    named += [("backward_grad_gather", v) for v in pick("k_grad_gather", 1)]
    for name, v in named:
        out[name].update(v)
print(json.dumps(out, indent=1))
