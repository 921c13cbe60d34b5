"""Diagnostic (EGR_TASK_TIMES=8 build): start / primary-step end / end of every task's BACKWARD chain and its number of primary hit rows."""
import importlib, sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, os.environ.get("VARIANT", "init"), seed=0); cam = syn.default_camera(); tg = syn.make_targets(W, H)
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
m.set_strands(1)
world = int(os.environ.get("EMU_WORLD", "1"))
if world > 1: m.set_partition(0, world); m.set_team_help(True)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **{k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()})
for _ in range(12):
    rt.zero_grad(); ren.render(camera, rt)
torch.cuda.synchronize()
st = m.get_stats()
tr = st.num_traversed_per_pixel.view(H, W); ac = st.num_accumulated_per_pixel.view(H, W)
t0 = tr[::8, 0::8].cpu().numpy().astype(np.int64).ravel(); t2 = ac[::8, 0::8].cpu().numpy().astype(np.int64).ravel()
t1 = tr[::8, 1::8].cpu().numpy().astype(np.int64).ravel(); rows = ac[::8, 1::8].cpu().numpy().astype(np.int64).ravel()
own = (t2 > t0) & (t1 >= t0)
t0, t1, t2, rows = t0[own], t1[own], t2[own], rows[own]
tot, prim, bounce = (t2 - t0) * 0.01, (t1 - t0) * 0.01, (t2 - t1) * 0.01  # (the primary step runs first, the bounce steps last)
o = np.argsort(-tot)[:10]
print("tasks", len(tot), "span us", (t2.max() - t0.min()) * 0.01, "sum / 3072 slots", tot.sum() / 3072, "mean task", tot.mean(), "max", tot.max())
print("heaviest tasks (us: total | bounce steps | primary step | primary hit rows):", [(round(float(tot[i]), 1), round(float(bounce[i]), 1), round(float(prim[i]), 1), int(rows[i])) for i in o])
print("mean: bounce steps", bounce.mean(), "primary step", prim.mean(), "primary rows", rows.mean(), "us per primary row", prim.sum() / max(rows.sum(), 1))
# what another ORDER of the same tasks would give (list schedule on the backward chain's wave slots): as started, longest first, and longest
# first by a proxy the forward chain knows before the backward starts (primary hit rows)
import heapq
def sim(order, slots=3072):
    h = [0.0] * slots; heapq.heapify(h); fin = 0.0
    for i in order:
        x = heapq.heappop(h); heapq.heappush(h, x + tot[i]); fin = max(fin, x + tot[i])
    return fin
print("list schedule of these task times on 3072 slots: in start order", round(sim(np.argsort(t0)), 1), "| longest first", round(sim(np.argsort(-tot)), 1), "| most primary rows first", round(sim(np.argsort(-rows)), 1), "| sum / slots", round(tot.sum() / 3072, 1))
