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

# The suite pins team help OFF at creation (conftest.py: many tests assert bit equality between two launches), but the kernels the product ships and bench.py
# times are the TEAM builds (k_forward_chain<.., 16>, k_backward_chain<4>: egr_set_team_help(1), the library default). Every test that compares with the ORACLE
# with a tolerance runs twice: help off (single-wave workgroups) and the product default (teams; help across the CU's waves and, round 6, across the GPU).
BOTH_HELP_MODES = pytest.mark.parametrize("team_help", [False, True], ids=["help_off", "product_default_help_on"])


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


def grads_vs_oracle_listing_flipped_pixels(ren, rt, o, camera, tg, W, H, name, max_flipped=8, bar=1e-3, loose=5e-3):
    """The north-star gradient bar (max-rel-err < 1e-3 of each tensor's max-abs) with a SHOWN reason for anything above it.

    A bounce ray that differs by an ulp between the two implementations can meet one grazing candidate (alpha = alpha_threshold at
    the clip sphere) more or less; on a small image one such hit is 1-2e-3 of a tensor's maximum. Instead of loosening the bar:
      1. one grad launch on both sides -> nine gradient tensors; if all are < bar, done;
      2. otherwise LIST the pixels whose rays differ: composited hits per bounce step (HIP: egr_debug_get_step_hits, oracle:
         num_hits[step]), total transmittance of a step (a flipped candidate behind the last composited hit only shows there), or two
         consecutive composited hits within 4 ulps of each other in the oracle (their ORDER - hence their two weights - is decided by
         the last bits of t, where this build contracts fmas and the oracle does not; exact ties: by the list order, which is
         unspecified upstream);
      3. assert there are at most `max_flipped` of them, take exactly those PIXELS out ON BOTH SIDES (pixel masks: egr_debug_set_pixel_mask /
         Oracle.set_pixel_mask - the same whole-image launch otherwise) and assert < bar on everything else.
    Whatever is listed, NO tensor may be further than `loose` (5e-3) from the oracle on all pixels.
    Returns (per-tensor errors over all pixels, per-tensor errors without the listed pixels, listed pixels)."""
    m = rt.cuda_module
    run_grad(ren, rt, camera)
    k = int(m.get_metadata().total_num_calls)
    o.total_num_calls = k - 1
    ref = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    live = [key for key in GRAD_KEYS if np.abs(ref[key]).max() > 0]
    err_all = {key: float(np.abs(gr[key] - ref[key]).max() / np.abs(ref[key]).max()) for key in live}
    for key in GRAD_KEYS:  # a tensor the configuration switches off stays exactly zero on both sides
        if key not in live:
            assert float(np.abs(gr[key]).max()) == 0.0, key
    worst_all = max(err_all.values())
    assert worst_all < loose, (name, err_all)  # unconditional: a listing never excuses more than this
    if worst_all < bar:
        report(name, worst_grad=f"{worst_all:.1e}", flipped_pixels=0)
        return err_all, err_all, []
    hits_h = m.debug_step_hits().numpy()  # [3,H,W], this grad launch
    hits_o = ref["num_composited_per_step"]
    m.get_metadata().total_num_calls.fill_(k - 1)  # the same rays once more, as images
    with torch.no_grad():
        rt(camera)
    o.total_num_calls = k - 1
    img_o = o.raytrace(False)
    tt_h = hip_outputs(rt)["output_total_transmittance"][..., 0]
    why = {"hits": np.any(hits_h != hits_o, axis=0), "T_total": np.any(np.abs(tt_h - img_o["output_total_transmittance"][..., 0]) > 2e-5, axis=0),
           "near_tie": ref["num_depth_ties"] > 0}
    flipped = why["hits"] | why["T_total"] | why["near_tie"]
    ys, xs = np.nonzero(flipped)
    listing = [(int(x), int(y), "+".join(k for k in why if why[k][y, x]), hits_h[:, y, x].tolist(), hits_o[:, y, x].tolist()) for y, x in zip(ys, xs)]
    assert 0 < len(listing) <= max_flipped, (name, err_all, listing[:20], len(listing))
    keep = ~flipped  # exactly the listed PIXELS are taken out, on both sides
    o.set_pixel_mask(keep)
    o.total_num_calls = k - 1
    ref_rest = o.raytrace(True, targets=tg)
    o.set_pixel_mask(None)
    rt.zero_grad()
    m.get_gaussians().total_weight.zero_()
    m.debug_set_pixel_mask(torch.from_numpy(keep.astype(np.uint8)).cuda())  # (a masked pixel is a pixel outside the image for every kernel of the launch)
    try:
        m.get_metadata().total_num_calls.fill_(k - 1)
        ren.render(camera, rt)
    finally:
        m.debug_set_pixel_mask(torch.empty(0, dtype=torch.uint8))
    torch.cuda.synchronize()
    gr_rest = hip_grads(rt)
    err_rest = {key: float(np.abs(gr_rest[key] - ref_rest[key]).max() / np.abs(ref[key]).max()) for key in live}
    worst_rest = max(err_rest.values())
    report(name, worst_grad_all_pixels=f"{worst_all:.1e}", worst_grad_without_listed=f"{worst_rest:.1e}", flipped_pixels=listing)
    assert worst_rest < bar, (name, err_rest, listing)
    return err_all, err_rest, listing
