"""Diagnostic (needs a build with EGR_TASK_TIMES=<step>): distribution of per-task walk / composite time of one forward step."""
import importlib, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
world = int(os.environ.get("EMU_WORLD", "8"))
g = syn.make_scene(N, "trained", seed=0); cam = syn.default_camera(); pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000); m = rt.cuda_module
m.set_strands(1)
if world > 1: m.set_partition(0, world)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
for _ in range(20):
    with torch.no_grad(): rt(camera)
torch.cuda.synchronize()
st = m.get_stats()
walk = st.num_traversed_per_pixel.view(H, W)[::8, ::8].float() * 0.01  # us
comp = st.num_accumulated_per_pixel.view(H, W)[::8, ::8].float() * 0.01
hc = st.num_traversed_per_pixel.view(H // 8, 8, W // 8, 8).clone()
hc[:, 0, :, 0] = 0
hmax = hc.amax(dim=(1, 3)).float(); hsum = hc.sum(dim=(1, 3)).float()
own = walk > 0
q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=walk.device)
def qs(x):
    s = x.sort().values
    return [round(float(v), 1) for v in s[(q * (s.numel() - 1)).long()]] + [round(float(s[-1]), 1)]
print("tasks", int(own.sum()))
print("walk us  q50/90/99/99.9/max", qs(walk[own]), "mean", float(walk[own].mean()))
print("comp us  q50/90/99/99.9/max", qs(comp[own]), "mean", float(comp[own].mean()))
tot = (walk + comp)[own]
print("total us q50/90/99/99.9/max", qs(tot), "mean", float(tot.mean()))
top = tot.topk(10).indices
print("top-10 tasks: walk, comp, tile max Hc (all steps, 63 px), tile sum Hc")
for i in top.tolist():
    print("  ", round(float(walk[own][i]), 1), round(float(comp[own][i]), 1), float(hmax[own][i]), float(hsum[own][i]))
print("all-tile Hc: mean of tile max", float(hmax[own].mean()), "mean of tile sum", float(hsum[own].mean()))
