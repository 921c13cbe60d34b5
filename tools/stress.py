"""Robustness sweep on the GPU (not part of the test suite): sizes, anisotropy and depth complexity the bench scene does not have.
Prints time, counters and status per case; small cases are compared with the oracle."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
from oracle import oracle as orc


def run(name, g, W, H, cam, fwd=400_000_000, bwd=300_000_000, check=False, iters=3):
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=fwd, ppll_backward_size=bwd)
    m = rt.cuda_module
    assert m.check_bvh() == 0, m.last_error()
    tg = syn.make_targets(W, H)
    images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
    camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
    for _ in range(iters):
        rt.zero_grad(); ren.render(camera, rt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        rt.zero_grad(); ren.render(camera, rt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters * 1e3
    c = m.get_counters()
    gf = m.get_gaussians().grad_flat
    line = f"{name}: {dt:.2f} ms/iter, rays {c[0:3]}, Hc/ray {[round(c[3+i]/max(c[i],1),1) for i in range(3)]}, Kc/ray {[round(c[6+i]/max(c[i],1),1) for i in range(3)]}, status {c[11]}, depth {c[12]}, grads finite {bool(torch.isfinite(gf).all())}"
    if check:
        o = orc.Oracle(W, H); o.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o.set_config(jitter_primary_rays=0, **syn.TRAIN_LOSS_WEIGHTS); o.set_gaussians(g); o.update_bvh()
        m.get_config().jitter_primary_rays.fill_(False)
        m.get_metadata().total_num_calls.zero_(); o.total_num_calls = 0 if hasattr(o, "total_num_calls") else None
        with torch.no_grad():
            rt(camera)
        ref = o.raytrace(False)
        out = m.get_framebuffer().output_rgb.cpu().numpy()
        mse = float(np.mean((out[0] - ref["output_rgb"][0]) ** 2))
        line += f", primary PSNR vs oracle {150.0 if mse == 0 else 10 * np.log10(1 / mse):.1f} dB"
    print(line, flush=True)


cam = syn.default_camera()
run("room 100k @1080p", syn.make_scene(100_000, "trained", seed=0), 1920, 1080, cam)
run("room 3M @1080p", syn.make_scene(3_000_000, "trained", seed=0), 1920, 1080, cam)
g = syn.random_blob_scene(200_000, seed=1, extent=2.0, depth_range=(1.0, 8.0), scale_range=(0.002, 0.3))
run("anisotropic blobs 200k (scale 0.002..0.3) @720p", g, 1280, 720, syn.plus_x_camera())
g = syn.random_blob_scene(3000, seed=2, extent=1.0, depth_range=(1.0, 6.0), scale_range=(0.01, 0.4))
run("anisotropic blobs 3k @128x96 vs oracle", g, 128, 96, syn.plus_x_camera(), fwd=50_000_000, bwd=50_000_000, check=True)
g = syn.make_scene(300_000, "init", seed=5)  # opacity 0.1: rays see hundreds of hits
run("room 300k, init opacity 0.1 @720p", g, 1280, 720, cam)
