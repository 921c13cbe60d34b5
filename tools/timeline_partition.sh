cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl8 -o t -- python bench.py --steps 12 --warmup 30 --no-cpu-baseline --profile-steps 0 --emulate-world 8 > gpurun_out/tl8.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/tl8/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
pro=[i for i,r in enumerate(rows) if "k_prologue" in r["Kernel_Name"]]
a,b=pro[-3],pro[-2]
it=rows[a:b]
t0=int(it[0]["Start_Timestamp"]); span=(int(rows[b]["Start_Timestamp"])-t0)/1e6
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in it)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print(f"iteration span {span:.3f} ms, GPU busy (union of kernels) {busy/1e6:.3f} ms, idle {span-busy/1e6:.3f} ms, kernels {len(it)}")
prev_end=t0
for r in it:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print(f'{n[:40]:40s} q={r.get("Queue_Id","?"):>3s} start={(s-t0)/1e6:8.3f} dur={(e-s)/1e6:7.3f} gap_before={(s-prev_end)/1e6:7.3f}')
    prev_end=max(prev_end,e)
PY
rm -rf gpurun_out/tl8
