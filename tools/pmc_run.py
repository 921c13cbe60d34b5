"""One full-size launch of a bench workload for rocprofv3 --pmc passes (kept minimal: PMC runs are slow).
PMC_CONFIG = C (1M gaussians, one training iteration; default) | B (100k, forward only); PMC_VARIANT = init (default) | trained."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
CONFIG, VARIANT = os.environ.get("PMC_CONFIG", "C"), os.environ.get("PMC_VARIANT", "init")
W, H, N = int(os.environ.get("PMC_W", 1920)), int(os.environ.get("PMC_H", 1080)), (100_000 if CONFIG == "B" else 1_000_000)
g = syn.make_scene(N, VARIANT, seed=0); cam = syn.default_camera(); tg = syn.make_targets(W, H)
pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000)
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
for _ in range(int(os.environ.get("ITERS", 1))):
    if CONFIG == "B":
        with torch.no_grad():
            rt(camera)
    else:
        rt.zero_grad()
        ren.render(camera, rt)
torch.cuda.synchronize()
print("counters", rt.cuda_module.get_counters())
