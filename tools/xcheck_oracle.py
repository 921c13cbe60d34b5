"""Mid-size cross-check against the CPU oracle (1M gaussians, 480x270, jitter off): per-pixel T_total is order independent, so any
difference beyond round-off means the two implementations accepted different candidate sets."""
import importlib, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
from oracle import oracle as orc
W, H, N = 480, 270, int(os.environ.get("XN", 1_000_000))
g = syn.make_scene(N, "trained", seed=0); cam = syn.default_camera()
pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=100_000_000, ppll_backward_size=100_000_000); m = rt.cuda_module
m.get_config().jitter_primary_rays.fill_(False)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
with torch.no_grad(): rt(camera)
fb = m.get_framebuffer(); st = m.get_stats()
o = orc.Oracle(W, H, threads=len(os.sched_getaffinity(0))); o.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o.set_config(jitter_primary_rays=0); o.set_gaussians(g); o.update_bvh()
t0 = time.time(); ref = o.raytrace(False); print("oracle s", round(time.time() - t0, 1), "status", m.get_counters()[11])
for k in ("output_total_transmittance", "output_transmittance", "output_rgb", "output_ray_direction"):
    a, b = getattr(fb, k).cpu().numpy(), ref[k]
    d = np.abs(a - b).reshape(3, H * W, -1).max(-1)
    print(k, "per step: max diff", [float(x) for x in d.max(1)], "pixels > 1e-4", [int(x) for x in (d > 1e-4).sum(1)])
ha = st.num_accumulated_per_pixel.cpu().numpy().reshape(-1)
print("num_accumulated differs in", int((ha != ref["num_accumulated"].reshape(-1)).sum()), "pixels of", H * W)
