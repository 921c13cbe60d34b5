import importlib, sys, os, numpy as np, torch
sys.path.insert(0, ".")
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
from oracle import oracle as orc
cam = syn.default_camera()
def run(n, W, H, bounces=2):
    g = syn.make_scene(max(n, 1), "trained", seed=3)
    g = {k: v[:n] for k, v in g.items()}
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=4_000_000, ppll_backward_size=4_000_000)
    m = rt.cuda_module; m.get_config().num_bounces.fill_(bounces)
    tg = syn.make_targets(W, H)
    images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
    camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
    with torch.no_grad(): rt(camera)
    out = m.get_framebuffer().output_rgb.cpu().numpy().copy()
    rt.zero_grad(); ren.render(camera, rt); torch.cuda.synchronize()
    c = m.get_counters(); gf = m.get_gaussians().grad_flat
    line = f"N={n} {W}x{H} b={bounces}: status {c[11]}, rays {c[0:3]}, finite out {np.isfinite(out).all()}, finite grads {bool(torch.isfinite(gf).all())}"
    if n > 0:
        o = orc.Oracle(W, H); o.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o.set_config(num_bounces=bounces, **syn.TRAIN_LOSS_WEIGHTS); o.set_gaussians(g); o.update_bvh()
        m.get_metadata().total_num_calls.zero_()
        with torch.no_grad(): rt(camera)
        out = m.get_framebuffer().output_rgb.cpu().numpy()
        ref = o.raytrace(False)
        mse = float(np.mean((out - ref["output_rgb"]) ** 2)); line += f", PSNR vs oracle {150.0 if mse == 0 else 10*np.log10(1/mse):.1f} dB"
    print(line, flush=True)
for n, W, H in ((1, 8, 8), (7, 17, 3), (9, 1, 1), (64, 33, 65), (500, 16, 16), (500, 128, 8)):
    try: run(n, W, H)
    except Exception as e: print(f"N={n} {W}x{H}: EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
try: run(0, 16, 16)
except Exception as e: print(f"N=0: EXCEPTION {type(e).__name__}: {str(e)[:200]}", flush=True)
