"""Where the caller-side (non-kernel) part of one training iteration goes: GPU time of each piece of GaussianRaytracer.__call__, HIP events, 50 reps."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic"); ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
W, H, N = 1920, 1080, 1_000_000
g = syn.make_scene(N, "trained", seed=0); cam = syn.default_camera(); tg = syn.make_targets(W, H)
rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000)
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
fb = rt.cuda_module.get_framebuffer()
def timed(name, f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    print(f"{name:28s} {a.elapsed_time(b) / reps:.4f} ms", flush=True)
timed("zero_grad", rt.zero_grad)
timed("export", rt._export_param_values)
timed("import", rt._import_param_gradients)
def targets():
    for n, k in (("target_diffuse", "diffuse_image"), ("target_specular", "specular_image"), ("target_depth", "depth_image"), ("target_normal", "normal_image"), ("target_roughness", "roughness_image"), ("target_f0", "f0_image")):
        getattr(fb, n).copy_(images[k].moveaxis(0, -1))
timed("6 target copies CHW->HWC", targets)
def cam_setup():
    R = camera.R.cuda(); Rb = rt.blender_rotation(R.clone()); c = rt.cuda_module.get_camera()
    c.znear.fill_(0.01); c.zfar.fill_(999.9); c.vertical_fov_radians.fill_(float(camera.FoVy)); c.set_pose(camera.camera_center.contiguous(), Rb.contiguous())
timed("camera setup", cam_setup)
timed("update_bvh", rt.cuda_module.update_bvh)
def full():
    rt.zero_grad(); ren.render(camera, rt)
timed("whole iteration (render)", full)
def call_only():
    rt.zero_grad(); rt(camera, target_diffuse=images["diffuse_image"], target_specular=images["specular_image"], target_depth=images["depth_image"], target_normal=images["normal_image"], target_roughness=images["roughness_image"], target_f0=images["f0_image"])
timed("whole iteration (__call__)", call_only)
