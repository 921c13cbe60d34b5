"""Per-launch duration of the forward chain over N consecutive training iterations (HIP events on the launch stream, no profiler):
looks for periodic slow launches. Usage: python tools/launch_times.py [N] [variant]"""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
n, variant = int(sys.argv[1]) if len(sys.argv) > 1 else 300, sys.argv[2] if len(sys.argv) > 2 else "init"
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, variant, seed=0); cam = syn.default_camera(); tg = syn.make_targets(W, H)
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **{k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()})
m.set_strands(1); m.enable_timing(True)
fw = []
for i in range(n):
    rt.zero_grad(); ren.render(camera, rt); torch.cuda.synchronize()
    fw.append(dict(m.last_kernel_ms())["forward_chain"])
x = np.array(fw); med = np.median(x[50:])
slow = np.nonzero(x > 1.2 * med)[0]
print(f"{variant}: {n} launches, forward chain median {med:.3f} ms, mean {x[50:].mean():.3f}, max {x.max():.3f}; launches > 1.2 x median: {slow.tolist()} -> {np.round(x[slow], 2).tolist()}")
