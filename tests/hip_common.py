"""Shared helpers of the GPU parity tests (HIP path through torch.classes.raytracer -> libraytracer.so -> C ABI vs the CPU oracle)."""
import importlib
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

PKG = "editable-gaussian-reflections_amd"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
OUT_KEYS = ["output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance",
            "output_total_transmittance", "output_ray_origin", "output_ray_direction", "output_final"]
GRAD_KEYS = ["dL_drgb", "dL_dnormal", "dL_df0", "dL_droughness", "dL_dopacity", "dL_dscale", "dL_dmean", "dL_drotation", "total_weight"]


@pytest.fixture(scope="module")
def ren():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    return importlib.import_module(PKG + ".renderer")


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 150.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)


def make_pair(ren, orc, g, cam, W, H, cfg=None, fwd=8_000_000, bwd=8_000_000, **kw):
    """Returns (GaussianRaytracer, Oracle) fed the same scene / camera / config."""
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=fwd, ppll_backward_size=bwd, **kw)
    o = orc.Oracle(W, H)
    o.set_camera(cam["origin"], cam["c2w"], cam["fov"], cam.get("znear", 0.01), cam.get("zfar", 999.9))
    o.set_gaussians(g)
    c = dict(loss_weight_diffuse=5.0, loss_weight_specular=3.0, loss_weight_normal=2.5, loss_weight_depth=2.5, loss_weight_f0=1.0,
             loss_weight_roughness=1.0)
    c.update(cfg or {})
    o.set_config(**c)
    mc = rt.cuda_module.get_config()
    for k, v in (cfg or {}).items():
        getattr(mc, k).fill_(v)
    o.update_bvh()
    return rt, o


def cam_obj(ren, cam, targets=None):
    images = {}
    if targets:
        images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in targets.items()}
    return ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)


def hip_outputs(rt):
    fb = rt.cuda_module.get_framebuffer()
    return {k: getattr(fb, k).cpu().numpy() for k in OUT_KEYS}


def hip_grads(rt):
    g = rt.cuda_module.get_gaussians()
    return {k: getattr(g, k).cpu().numpy() for k in GRAD_KEYS}


def run_grad(ren, rt, camera):
    rt.zero_grad()
    rt.cuda_module.get_gaussians().total_weight.zero_()
    ren.render(camera, rt)
    torch.cuda.synchronize()


def report(name, **kv):
    """Measured levels go to stdout (pytest -s / the captured log) so that the asserted bars can be kept at what is achieved."""
    print("REPORT " + name + ": " + ", ".join(f"{k}={v}" for k, v in kv.items()), flush=True)


def mismatch_list(a, b, limit=12):
    idx = np.flatnonzero(np.asarray(a).reshape(-1) != np.asarray(b).reshape(-1))
    return [(int(i), int(np.asarray(a).reshape(-1)[i]), int(np.asarray(b).reshape(-1)[i])) for i in idx[:limit]], int(idx.size)




def generic_targets(syn, W, H):
    """syn.make_targets with roughness / f0 moved off the scene's own wall values (0.1 / 0.04): where a fully opaque wall renders
    exactly its target, sign(output - target) is decided by the last bit of rounding, i.e. by noise in BOTH implementations."""
    tg = syn.make_targets(W, H)
    tg["roughness"] = tg["roughness"] + np.float32(0.23)
    tg["f0"] = tg["f0"] + np.float32(0.17)
    return tg
