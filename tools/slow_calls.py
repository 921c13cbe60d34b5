"""Which launch (call number = jitter / bounce seed) is slow, and what is different in it."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
variant = sys.argv[1] if len(sys.argv) > 1 else "init"
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, variant, seed=0); cam = syn.default_camera(); tg = syn.make_targets(W, H)
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **{k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()})
m.set_strands(1); m.enable_timing(True)
for _ in range(30):
    rt.zero_grad(); ren.render(camera, rt)
torch.cuda.synchronize()
for call in (9, 10, 11, 30):
    m.get_metadata().total_num_calls.fill_(call - 1)
    with torch.no_grad(): rt(camera)
    torch.cuda.synchronize()
    ms = dict(m.last_kernel_ms()); tr = m.get_stats().num_traversed_per_pixel.cpu().numpy()
    print(f"call {call} NO-GRAD: forward {ms['forward_chain']:.2f} ms; max evaluated per pixel {int(tr.max())}", flush=True)
    m.get_metadata().total_num_calls.fill_(call - 1)
    with torch.no_grad(): rt(camera, force_update_bvh=True)
    torch.cuda.synchronize()
    ms = dict(m.last_kernel_ms()); tr = m.get_stats().num_traversed_per_pixel.cpu().numpy()
    print(f"call {call} NO-GRAD after a refit: forward {ms['forward_chain']:.2f} ms; max evaluated per pixel {int(tr.max())}", flush=True)
    m.get_metadata().total_num_calls.fill_(call - 1)
    rt.zero_grad(); ren.render(camera, rt); torch.cuda.synchronize()
    ms = dict(m.last_kernel_ms()); c = m.get_counters()
    hits = m.debug_step_hits().numpy()
    tr = m.get_stats().num_traversed_per_pixel.cpu().numpy()
    y, x = np.unravel_index(int(hits.sum(0).argmax()), hits.shape[1:])
    y2, x2 = np.unravel_index(int(tr.argmax()), tr.shape)
    print(f"call {call}: forward {ms['forward_chain']:.2f} ms backward {ms['backward_chain']:.2f}; status {c[11]} ext blocks {c[17]} arena blocks {c[15]}; max composited per pixel (all steps) {int(hits.sum(0).max())} at ({x},{y}) per step {hits[:, y, x].tolist()}; max evaluated per pixel {int(tr.max())} at ({x2},{y2}); accepted {c[19:22]}", flush=True)
