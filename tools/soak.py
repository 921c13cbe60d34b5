"""Soak run on the GPU (not part of the test suite): full-size training iterations from MOVING cameras with the fused host
step and periodic rebuilds - every iteration checks the status word; finiteness, BVH validity and the image error every 50.
    python tools/soak.py [iterations]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
tr = importlib.import_module("editable-gaussian-reflections_amd.trainer")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, "trained", seed=0); base = syn.default_camera(); tg = syn.make_targets(W, H)
pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
lrs = dict(xyz=1.6e-5, normal=2e-3, roughness=2e-3, f0=2e-3, f_dc=2e-3, opacity=5e-3, scaling=1e-3, rotation=1e-3)
step = tr.FusedTrainStep(pc, rt, lrs, scale_decay=1.0, xyz_schedule=dict(lr_init=1.6e-5, lr_final=1.6e-7, lr_delay_mult=0.01, max_steps=30000))
rng = np.random.default_rng(1)
def camera(i):  # orbit + jitter around the default pose
    a = 0.35 * np.sin(i * 0.37) + 0.05 * rng.standard_normal()
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    o = R @ np.asarray(base["origin"], np.float64) + 0.05 * rng.standard_normal(3)
    return ren.camera_from_c2w(o.astype(np.float32), (R @ np.asarray(base["c2w"], np.float64)).astype(np.float32), base["fov"], **images)
t0 = time.perf_counter(); bad = 0
for it in range(1, iters + 1):
    step.update_learning_rate(it)
    ren.render(camera(it), rt)
    step.step()
    if it % 125 == 0:
        rt.rebuild_bvh()
    c = m.get_counters()  # synchronises
    if c[11] != 0:
        bad += 1; print(f"iteration {it}: status {c[11]}", flush=True)
    if it % 50 == 0:
        finite = all(bool(torch.isfinite(p).all()) for p in pc.parameters())
        print(f"iteration {it}: {1e3 * (time.perf_counter() - t0) / it:.1f} ms/iter incl. sync, rays {list(c[0:3])}, records {c[13]}, params finite {finite}, bvh ok {m.check_bvh() == 0}", flush=True)
        assert finite and m.check_bvh() == 0
print("done:", iters, "iterations,", bad, "with a non-zero status")
