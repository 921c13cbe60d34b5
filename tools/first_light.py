"""Dev diagnostic: HIP path vs CPU oracle on a small synthetic scene. Run on the GPU box."""
import importlib, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("editable-gaussian-reflections_amd")
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
from oracle import oracle as orc

W, H, N = int(os.environ.get("W", 64)), int(os.environ.get("H", 64)), int(os.environ.get("N", 2000))
nb = int(os.environ.get("NB", 2))
g = syn.make_scene(N, "trained", seed=0)
cam = syn.default_camera()
print("torch", torch.__version__, torch.cuda.get_device_name(0))
pc = ren.GaussianParams(g)
rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=20_000_000, ppll_backward_size=20_000_000)
m = rt.cuda_module
print("bvh check", m.check_bvh(), m.last_error())
cfg = m.get_config(); cfg.jitter_primary_rays.fill_(False); cfg.num_bounces.fill_(nb)
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"])
with torch.no_grad():
    rt(camera)
torch.cuda.synchronize()
print("counters", m.get_counters())
fb = m.get_framebuffer()
o = orc.Oracle(W, H); o.set_camera(cam["origin"], cam["c2w"], cam["fov"]); o.set_gaussians(g); o.update_bvh()
o.set_config(jitter_primary_rays=0, num_bounces=nb, **syn.TRAIN_LOSS_WEIGHTS)
ref = o.raytrace(False)
def psnr(a, b):
    mse = np.mean((a - b) ** 2); return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)
for k in ["output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance", "output_total_transmittance", "output_ray_origin", "output_ray_direction", "output_final"]:
    a = getattr(fb, k).cpu().numpy().astype(np.float64); b = ref[k]
    for s in range(a.shape[0]):
        print(f"{k}[{s}] maxabs {np.abs(a[s]-b[s]).max():.3e} psnr {psnr(a[s], b[s]):.1f}")
st = m.get_stats()
ht = st.num_traversed_per_pixel.cpu().numpy(); ha = st.num_accumulated_per_pixel.cpu().numpy()
print("Hc mismatch px", (ht != ref["num_traversed"]).sum(), "of", ht.size, "Kc(last) mismatch", (ha != ref["num_accumulated"]).sum())
print("seeds equal", np.array_equal(m.get_metadata().random_seeds.cpu().numpy().astype(np.uint32).reshape(H, W), ref["random_seeds"].reshape(H, W)))
# gradients
tg = syn.make_targets(W, H)
camera_t = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **{k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0) for k, v in tg.items()})
rt.zero_grad(); m.get_gaussians().total_weight.zero_()
ren.render(camera_t, rt)
torch.cuda.synchronize()
print("counters(grad)", m.get_counters())
o.total_num_calls = int(m.get_metadata().total_num_calls.item()) - 1
refg = o.raytrace(True, targets=tg)
gg = m.get_gaussians()
for k in ["dL_drgb", "dL_dnormal", "dL_df0", "dL_droughness", "dL_dopacity", "dL_dscale", "dL_dmean", "dL_drotation", "total_weight"]:
    a = getattr(gg, k).cpu().numpy().astype(np.float64); b = refg[k]
    print(f"{k}: max|ref| {np.abs(b).max():.3e} maxabs err {np.abs(a-b).max():.3e} rel-to-max {np.abs(a-b).max()/(np.abs(b).max()+1e-30):.3e}")
