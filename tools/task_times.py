"""Diagnostic (needs a build with EGR_TASK_TIMES=<step>): start / end time of every task of one forward step -> how full the
wave slots are over the kernel's lifetime, and what a longest-first order of the same tasks would give."""
import importlib, sys, os, torch, numpy as np, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
world = int(os.environ.get("EMU_WORLD", "1"))
g = syn.make_scene(N, os.environ.get("VARIANT", "trained"), seed=0); cam = syn.default_camera(); pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
m.set_strands(1)
m.get_config().num_bounces.fill_(int(os.environ["TT_STEP"]))  # the measured step must be the last one (later steps overwrite the stats)
if world > 1: m.set_partition(0, world)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
for _ in range(20):
    with torch.no_grad(): rt(camera)
if os.environ.get("CALL"):  # the launch with this call number (= jitter / bounce seeds) is the one measured
    m.get_metadata().total_num_calls.fill_(int(os.environ["CALL"]) - 1)
    with torch.no_grad(): rt(camera)
torch.cuda.synchronize()
st = m.get_stats()
RY = 4 if int(os.environ.get("EGR_RAYS_PER_TASK", "64")) == 32 else 8  # task height in pixels (8x8 tasks, or 8x4 with EGR_RAYS_PER_TASK=32)
t0 = st.num_traversed_per_pixel.view(H, W)[::RY, ::8].cpu().numpy().astype(np.int64).ravel()
t1 = st.num_accumulated_per_pixel.view(H, W)[::RY, ::8].cpu().numpy().astype(np.int64).ravel()
tm = st.num_traversed_per_pixel.view(H, W)[::RY, 1::8].cpu().numpy().astype(np.int64).ravel()
own = t1 > t0
def px(img, k): return img.view(H, W)[::RY, k::8].cpu().numpy().astype(np.int64).ravel()[own]
extra = {"walk batches": px(st.num_traversed_per_pixel, 2), "evaluation batches": px(st.num_accumulated_per_pixel, 2), "offers": px(st.num_traversed_per_pixel, 3), "tall-stack batches": px(st.num_accumulated_per_pixel, 3),
         "us waiting for helpers (helping meanwhile)": px(st.num_traversed_per_pixel, 4) * 0.01, "longest list": px(st.num_accumulated_per_pixel, 4)}
t0, t1, tm = t0[own], t1[own], tm[own]
walk, sel = (tm - t0) * 0.01, (t1 - tm) * 0.01
order = np.argsort(-(t1 - t0))[:12]
print("heaviest tasks (us): total / walk / selection+compositing:", [(round(float((t1 - t0)[i]) * 0.01, 1), round(float(walk[i]), 1), round(float(sel[i]), 1)) for i in order])
for k, a in extra.items(): print(f"   {k}: heaviest tasks", [round(float(a[i]), 1) for i in order], "| mean over all tasks", round(float(a.mean()), 1))
print("all tasks: walk share of the task time: mean", float(walk.sum() / (walk.sum() + sel.sum())))
base = t0.min()
s, e = (t0 - base) * 0.01, (t1 - base) * 0.01  # us
dur = e - s
slots = int(os.environ.get("SLOTS", 4096))  # resident waves of the forward chain (16 per CU x 256 CUs)
print("tasks", len(s), "kernel span us", e.max(), f"sum of task times / {slots} slots", dur.sum() / slots, "max task", dur.max(), "mean", dur.mean())
# occupancy over time
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    T = e.max() * frac
    print(f"  at {T:8.1f} us: running {int(((s <= T) & (e > T)).sum()):5d}  finished {int((e <= T).sum()):6d}")
last_start = s.max()
print("last task starts at", last_start, "-> tail", e.max() - last_start)
def simulate(order):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + dur[i]); end = max(end, t + dur[i])
    return end
# what intra-tile parallelism could buy: every task split over k waves without loss (k x the tasks at 1/k of the time each), and the
# same with only the walk / evaluation part split (selection + compositing stay on one wave per ray)
for k in (2, 4):
    def sim(durs):
        h = [0.0] * slots; heapq.heapify(h); end = 0.0
        for dd in sorted(durs, reverse=True):
            t = heapq.heappop(h); heapq.heappush(h, t + dd); end = max(end, t + dd)
        return end
    both = sim(np.repeat(dur / k, k))
    print(f"bound with every task split over {k} waves: {both:.1f} us (longest first); heaviest task / {k}: {dur.max() / k:.1f} us; walk split only, heaviest: {(walk / k + sel).max():.1f} us")
print("list schedule, same durations: start order", simulate(np.argsort(s)), " longest first", simulate(np.argsort(-dur)), " shortest first", simulate(np.argsort(dur)))
full = np.zeros(own.shape, np.float64); full[own] = dur  # per tile, raster order (0 = not this rank's)
np.save(os.path.join(ROOT, "gpurun_out", f"task_dur_w{world}_step{os.environ['TT_STEP']}.npy"), full)
