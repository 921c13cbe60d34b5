"""Diagnostic (EGR_TASK_TIMES=9 build): per-task time of the whole forward chain and of each step, for one call number."""
import importlib, sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, os.environ.get("VARIANT", "init"), seed=0); cam = syn.default_camera()
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
m.set_strands(1)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
for _ in range(20):
    with torch.no_grad(): rt(camera)
for call in [int(x) for x in os.environ.get("CALLS", "10,11").split(",")]:
    m.get_metadata().total_num_calls.fill_(call - 1)
    with torch.no_grad(): rt(camera)
    torch.cuda.synchronize()
    t = m.get_stats().num_traversed_per_pixel.view(H // 8, 8, W // 8, 8)[:, 0, :, :4].cpu().numpy().astype(np.int64).reshape(-1, 4) * 0.01  # us
    d = np.diff(t, axis=1)  # per step
    tot = t[:, 3] - t[:, 0]
    ok = (d >= 0).all(1)
    o = np.argsort(-np.where(ok, tot, 0))[:6]
    print(f"call {call}: heaviest chains (us: total | step 0, 1, 2):", [(round(float(tot[i]), 1), [round(float(x), 1) for x in d[i]], "tile", int(i % (W // 8)), int(i // (W // 8))) for i in o], "mean", round(float(tot[ok].mean()), 1), flush=True)
