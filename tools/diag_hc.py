import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
z = np.load(os.path.join(ROOT, "tests/golden/scene_2k_64.npz"))
W, H = int(z["W"]), int(z["H"])
g = {k[2:]: z[k] for k in z.files if k.startswith("g_")}; cam = {k[4:]: z[k] for k in z.files if k.startswith("cam_")}
for nb in (0, 1, 2):
    pc = ren.GaussianParams(g); rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000)
    rt.cuda_module.get_config().num_bounces.fill_(nb)
    with torch.no_grad(): rt(ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"]))
    ht = rt.cuda_module.get_stats().num_traversed_per_pixel.cpu().numpy()
    from oracle import oracle as orc
    o = orc.Oracle(W, H); o.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o.set_gaussians(g); o.update_bvh(); o.set_config(num_bounces=nb)
    ref = o.raytrace(False)["num_traversed"]
    d = ht - ref
    print("bounces", nb, "mean hip", ht.mean(), "mean ref", ref.mean(), "frac hip>ref", (d > 0).mean(), "max excess", d.max(), "min", d.min())
