"""Diagnostic (EGR_TASK_TIMES=9 build): per-task time of the whole forward chain and of each step, for one call number."""
import importlib, sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, os.environ.get("VARIANT", "init"), seed=0); cam = syn.default_camera()
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
m.set_strands(1)
world = int(os.environ.get("EMU_WORLD", "1"))
if world > 1: m.set_partition(int(os.environ.get("EMU_RANK", "0")), world)  # (build / run with EGR_RAYS_PER_TASK=64: the stamps sit in the first pixels of 8x8 tiles)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
for _ in range(20):
    with torch.no_grad(): rt(camera)
for call in [int(x) for x in os.environ.get("CALLS", "10,11").split(",")]:
    m.get_metadata().total_num_calls.fill_(call - 1)
    with torch.no_grad(): rt(camera)
    torch.cuda.synchronize()
    raw = m.get_stats().num_traversed_per_pixel.view(H // 8, 8, W // 8, 8)[:, 0, :, :5].cpu().numpy().astype(np.int64).reshape(-1, 5)
    t, leaves = raw[:, :4] * 0.01, raw[:, 4]  # us; leaves the tile's primary step evaluated
    d = np.diff(t, axis=1)  # per step
    tot = t[:, 3] - t[:, 0]
    ok = (d >= 0).all(1) & (t[:, 0] > 0) & (tot > 0)  # (tiles of other ranks carry no stamps)
    o = np.argsort(-np.where(ok, tot, 0))[:6]
    print(f"call {call}: heaviest chains (us: total | step 0, 1, 2):", [(round(float(tot[i]), 1), [round(float(x), 1) for x in d[i]], "tile", int(i % (W // 8)), int(i // (W // 8))) for i in o], "mean", round(float(tot[ok].mean()), 1), "mean per step", [round(float(x), 1) for x in d[ok].mean(0)], "tiles", int(ok.sum()), flush=True)
    # how full the wave slots are over the kernel, and what other tile orders of the SAME chain durations would give (list scheduling)
    import heapq
    start, end = t[ok, 0], t[ok, 3]
    base = start.min(); s, e = start - base, end - base; dur = e - s
    slots = int(os.environ.get("SLOTS", 4096))
    def sim(order):
        h = [0.0] * slots; heapq.heapify(h); fin = 0.0
        for i in order:
            x = heapq.heappop(h); heapq.heappush(h, x + dur[i]); fin = max(fin, x + dur[i])
        return fin
    # longest first inside each of the 8 XCD chunks only (compact image blocks stay together): chunks = 4 x 2 blocks of the image
    tx, ty = np.arange(len(ok))[ok] % (W // 8), np.arange(len(ok))[ok] // (W // 8)
    blk = np.minimum(3, tx * 4 // (W // 8)) + 4 * np.minimum(1, ty * 2 // (H // 8))
    per_chunk = [np.flatnonzero(blk == b)[np.argsort(-dur[blk == b])] for b in range(8)]
    inter = [int(x) for tup in zip(*[list(p) + [-1] * (max(map(len, per_chunk)) - len(p)) for p in per_chunk]) for x in tup if x >= 0]
    lv = leaves[ok]
    print(f"   leaves per tile mean {lv.mean():.1f} max {lv.max()}; correlation of the chain's duration with the leaf count {np.corrcoef(lv, dur)[0, 1]:.3f}, of the primary step's {np.corrcoef(lv, d[ok][:, 0])[0, 1]:.3f}; list schedule in descending LEAF order {sim(np.argsort(-lv)):.0f} us", flush=True)
    nat = np.argsort(s)
    for kk in (1.25, 1.5, 2.0, 3.0):
        heavy = lv > kk * lv.mean()
        order = list(np.flatnonzero(heavy)[np.argsort(-lv[heavy])]) + [i for i in nat if not heavy[i]]
        print(f"   tiles with more than {kk} x the mean leaf count first (descending, {int(heavy.sum())} tiles), the rest in start order: {sim(order):.0f} us", flush=True)
    np.save(os.environ.get("CHAIN_DUMP", "/tmp/chain_dump") + f"_{os.environ.get('VARIANT', 'init')}.npy", np.stack([lv, dur, d[ok][:, 0], d[ok][:, 1], d[ok][:, 2], s]))
    print(f"   span {e.max():.0f} us | sum / {slots} slots {dur.sum() / slots:.0f} | list schedule: start order {sim(np.argsort(s)):.0f}, longest first {sim(np.argsort(-dur)):.0f}, longest first inside each XCD block {sim(inter):.0f} | last start {s.max():.0f}", flush=True)
