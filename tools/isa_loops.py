"""Static instruction budget of the hot loops of a kernel, from the compiler's assembly (works without a GPU):
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o trace.s csrc/trace.hip;  python tools/isa_loops.py trace.s <kernel name substring>
For every loop (LLVM's "Loop Header" / "in Loop: Header=" block comments) the instructions of the blocks whose INNERMOST loop it is, by unit (VALU / SALU /
LDS / vector memory / scalar memory / branch), and the mnemonics that identify it (s_load_dwordx16 = the primary tiles' leaf loop, v_med3 = the selection
scan, v_mbcnt = a walk's compaction ...). A loop body with branches is counted whole: the executed path is shorter."""
import re, sys, collections
path, want = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_forward_chainILb1ELb0ELi16"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cur, parent, depth = None, {}, {}
count = collections.defaultdict(collections.Counter)
sig = collections.defaultdict(collections.Counter)
def unit(m):
    if m.startswith(("v_", "V_")): return "VALU"
    if m.startswith("ds_"): return "LDS"
    if m.startswith(("global_", "buffer_", "scratch_", "flat_")): return "VMEM" if not m.startswith("scratch_") else "SCRATCH"
    if m.startswith(("s_load", "s_buffer_load", "s_store")): return "SMEM"
    if m.startswith(("s_cbranch", "s_branch")): return "BRANCH"
    if m.startswith("s_waitcnt") or m.startswith("s_nop"): return "WAIT"
    return "SALU"
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        c = m.group(2) or ""
        h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
        if "Loop Header" in c and "in Loop" not in c.split("=>")[0]:
            cur = m.group(1)[2:]; d = re.search(r"Depth=(\d+)", c); depth[cur] = int(d.group(1)) if d else 0
        elif "Loop Header" in c:  # an inner loop's header inside an outer loop: the header line names both
            cur = m.group(1)[2:]; d = re.findall(r"Depth=(\d+)", c); depth[cur] = int(d[-1]) if d else 0
        elif h:
            cur = h.group(1); depth.setdefault(cur, int(h.group(2)))
        else:
            cur = None
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")): continue
    mn = t.split()[0]
    if cur is not None:
        count[cur][unit(mn)] += 1
        if mn in ("s_load_dwordx16", "v_med3_f32", "v_exp_f32", "v_sqrt_f32", "v_rcp_f32", "ds_bpermute_b32", "v_mbcnt_hi_u32_b32", "global_atomic_add_u32", "s_memrealtime", "v_cvt_f32_u32_e32", "global_load_dwordx4", "s_load_dwordx4", "v_readlane_b32", "v_writelane_b32", "scratch_load_dword", "scratch_store_dword"):
            sig[cur][mn] += 1
rows = sorted(count.items(), key=lambda kv: -sum(kv[1].values()))
print(f"{'loop':12s} {'depth':5s} {'total':>6s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'SMEM':>5s} {'BR':>4s} {'WAIT':>5s} {'SCR':>4s}  signature")
for k, c in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{k:12s} {depth.get(k, 0):5d} {sum(c.values()):6d} {c['VALU']:6d} {c['SALU']:6d} {c['LDS']:5d} {c['VMEM']:5d} {c['SMEM']:5d} {c['BRANCH']:4d} {c['WAIT']:5d} {c['SCRATCH']:4d}  " + ", ".join(f"{a} x{b}" for a, b in sig[k].most_common(6)))
