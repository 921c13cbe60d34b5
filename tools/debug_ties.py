import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
from oracle import oracle as orc
from hip_common import make_pair, cam_obj, hip_outputs
W, H = 48, 32
copies = int(os.environ.get("COPIES", 20))
g = syn.make_scene(600, "init", seed=13)
gN = {k: np.concatenate([v] * copies, 0) for k, v in g.items()}
cam = syn.default_camera()
rt, o = make_pair(ren, orc, gN, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0, transmittance_threshold=0.0))
m = rt.cuda_module
print("check_bvh", m.check_bvh())
with torch.no_grad(): rt(cam_obj(ren, cam))
ref = o.raytrace(False)
ha = m.get_stats().num_accumulated_per_pixel.cpu().numpy().reshape(-1)
ra = ref["num_accumulated"].reshape(-1)
print("hip mod16 hist", np.bincount(ha % 16, minlength=16)); print("orc mod16 hist", np.bincount(ra % 16, minlength=16))
bad = np.flatnonzero(ha != ra)
print("nbad", bad.size, bad[:10], ha[bad[:10]], ra[bad[:10]])
# instance records of the copies: bit-identical?
M, Wm, A = [t.numpy() for t in m.debug_instances()]
n0 = 600
same = all(np.array_equal(Wm[:n0].view(np.uint32), Wm[c * n0:(c + 1) * n0].view(np.uint32)) for c in range(1, copies))
print("W records bit-identical across copies:", same)
# exact-stats mode: candidate counts vs oracle
m.set_exact_stats(True); m.get_metadata().total_num_calls.zero_()
with torch.no_grad(): rt(cam_obj(ren, cam), force_update_bvh=True)
ht = m.get_stats().num_traversed_per_pixel.cpu().numpy().reshape(-1)
print("traversed mismatch", int((ht != ref["num_traversed"].reshape(-1)).sum()), "hip mod", np.bincount(ht % copies, minlength=copies)[:4], "orc mod", np.bincount(ref["num_traversed"].reshape(-1) % copies, minlength=copies)[:4])
ha2 = m.get_stats().num_accumulated_per_pixel.cpu().numpy().reshape(-1)
print("exact mode nbad", int((ha2 != ra).sum()))
o2 = orc.Oracle(W, H, use_bvh=False); o2.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o2.set_gaussians(gN)
o2.set_config(jitter_primary_rays=0, num_bounces=0, transmittance_threshold=0.0); o2.update_bvh()
r2 = o2.raytrace(False)
print("oracle bvh vs brute nbad", int((r2["num_accumulated"].reshape(-1) != ra).sum()))
