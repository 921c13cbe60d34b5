"""GPU tests of the BASELINE.json configurations and of the caller-visible semantics the round-1 review found untested:

  config 1  model directory in the reference's on-disk layout (PLY + transforms json) -> 256x256 render vs the committed oracle image
  config 2  100k Gaussians, 1080p, forward only, measure_fps.py protocol (no_grad, no BVH update)
  config 3  1M Gaussians, 1080p, forward + backward - the dense-INIT cloud (the trained-like variant is in test_hip_parity.py)
  + global_scale_factor / exp_power other than the defaults, live scale / rotation in a grad launch without a refit,
    N = 0 and N = 1, tiny images, stress scenes, the fused host step after render().
"""
import importlib
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from hip_common import BOTH_HELP_MODES, GOLD, GRAD_KEYS, OUT_KEYS, PKG, cam_obj, generic_targets, grads_vs_oracle_listing_flipped_pixels, hip_grads, hip_outputs, make_pair, mismatch_list, psnr, ren, report, run_grad  # noqa: F401


# ------------------------------------------------------------------------------------------------ config 1
def test_config_a_model_directory_renders_the_golden_image(ren, syn):
    """BASELINE config 1 (synthetic substitute): point_cloud.ply in the layout of scene/gaussian_model.py:366-407 and
    a transforms_test.json frame go through formats.py and the caller mirror (Camera.R / camera_center / FoVy, pose flip,
    export, rebuild) into the HIP tracer at 256x256; the image is the one the CPU oracle rendered from the same files
    (tests/golden/make_config_a.py)."""
    fmt = importlib.import_module(PKG + ".formats")
    d = os.path.join(GOLD, "config_a")
    W = H = 256
    g = fmt.load_gaussians_ply(os.path.join(d, "point_cloud.ply"))
    cfg = fmt.load_cfg(os.path.join(d, "cfg.json"))
    fr = fmt.read_transforms(os.path.join(d, "transforms_test.json"), W, H)[0]
    assert g["mean"].shape == (3000, 3) and g["rotation"].shape == (3000, 4)
    pc = ren.GaussianParams(g, cfg=SimpleNamespace(**{k: v for k, v in cfg.items() if k not in ("znear", "zfar")}))
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=30_000_000, ppll_backward_size=30_000_000)
    rt.cuda_module.get_config().jitter_primary_rays.fill_(False)
    camera = SimpleNamespace(R=np.asarray(fr["R"], np.float32), FoVy=float(fr["FovY"]),  # what scene/cameras.py:22 holds for the frame
                             camera_center=torch.tensor(np.asarray(fr["c2w"][:3, 3], np.float32)).cuda())
    with torch.no_grad():
        pkg = ren.render(camera, rt, targets_available=False, znear=cfg["znear"], zfar=cfg["zfar"])
    z = np.load(os.path.join(d, "golden_256.npz"))
    final = pkg.final.cpu().numpy()  # [1,3,H,W] like gaussian_renderer.py:74-92
    assert final.shape == (1, 3, H, W) and pkg.rgb.shape == (3, 3, H, W)
    levels = dict(final=psnr(np.moveaxis(final, 1, -1), z["output_final"].astype(np.float32)),
                  rgb0=psnr(np.moveaxis(pkg.rgb.cpu().numpy()[0], 0, -1), z["output_rgb0"].astype(np.float32)),
                  depth0=psnr(np.moveaxis(pkg.depth.cpu().numpy()[0], 0, -1) / 4.0, z["output_depth0"].astype(np.float32) / 4.0))
    ha = rt.cuda_module.get_stats().num_accumulated_per_pixel.cpu().numpy()
    bad, nbad = mismatch_list(ha, z["num_accumulated"])
    report("config_a", **{k: round(v, 1) for k, v in levels.items()}, acc_mismatch=nbad, of=W * H)
    assert levels["final"] > 60 and levels["rgb0"] > 65 and levels["depth0"] > 60, levels  # fixture is float16: ~70 dB ceiling; bar 50
    assert nbad <= 0.01 * W * H  # hit count of the LAST step: bounce rays differ by ulps between CPU and GPU
    assert rt.cuda_module.get_counters()[11] == 0


# ------------------------------------------------------------------------------------------------ config 2
@BOTH_HELP_MODES
def test_config_b_100k_1080p_forward_only(ren, orc, syn, team_help):
    """BASELINE config 2: synthetic dense-init cloud, N = 100k, 1920x1080, forward only under no_grad with no BVH update between
    frames (measure_fps.py:27-52). Size-independent properties at full size + the oracle on the same scene at low resolution."""
    W, H, N = 1920, 1080, 100_000
    g = syn.make_scene(N, "init", seed=0)
    cam = syn.default_camera()
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=1_000_000, team_help=team_help)
    m = rt.cuda_module
    fb = m.get_framebuffer()
    with torch.no_grad():
        for _ in range(2):
            rt(cam_obj(ren, cam))
    c = m.get_counters()
    assert c[11] == 0 and c[0] == W * H and c[13] == 0
    st = m.get_stats()
    assert int(st.num_traversed_per_pixel.sum().item()) == c[3] + c[4] + c[5]
    T, Tt = fb.output_transmittance, fb.output_total_transmittance
    assert bool((Tt <= T + 4e-6).all()) and bool((T <= 1.0).all()) and bool((Tt >= 0).all())
    assert float((fb.output_final[0] - fb.output_rgb.sum(0)).abs().max()) < 1e-5
    assert bool(torch.isfinite(fb.output_final).all())
    a = fb.output_final.clone()
    m.get_metadata().total_num_calls.sub_(1)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    if team_help:  # with help only the ORDER of exactly tied depths of bounce rays depends on timing: a 1080p frame holds such a tie in ~ 0.1 % of its bounce rays
        assert float((a != fb.output_final).any(-1).float().mean()) < 2e-3
    else:
        assert torch.equal(a, fb.output_final)  # idempotent: same call counter, bit-identical image
    report("config_b", rays=list(c[0:3]), evaluated_per_ray=[round(c[3 + i] / max(c[i], 1), 1) for i in range(3)],
           composited_per_ray=[round(c[6 + i] / max(c[i], 1), 1) for i in range(3)])
    # the same frame against the oracle AT THE CONFIG'S OWN SIZE: the pixels of 48 macro tiles (pixel mask on both sides), reference defaults
    # (jitter on, two bounces), every output buffer and step
    crop = np.zeros((H, W), bool)
    for mx, my in CROP_TILES:
        crop[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] = True
    o = orc.Oracle(W, H)
    o.set_camera(cam["origin"], cam["c2w"], cam["fov"], cam.get("znear", 0.01), cam.get("zfar", 999.9))
    o.set_gaussians(g)
    o.set_config()
    o.update_bvh()
    o.set_pixel_mask(crop)
    o.total_num_calls = 6
    ref = o.raytrace(False)
    m.debug_set_pixel_mask(torch.from_numpy(crop.astype(np.uint8)).cuda())
    try:
        for name in OUT_KEYS:
            getattr(fb, name).zero_()
        m.get_metadata().total_num_calls.fill_(6)
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        assert m.get_counters()[0] == int(crop.sum())
        out = hip_outputs(rt)
    finally:
        m.debug_set_pixel_mask(torch.empty(0, dtype=torch.uint8))
    lv = {}
    for key in ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance", "output_total_transmittance"):
        for s_ in range(3):
            lv[f"{key}[{s_}]"] = round(float(psnr(out[key][s_][crop], ref[key][s_][crop])), 1)
    lv["output_final"] = round(float(psnr(out["output_final"][0][crop], ref["output_final"][0][crop])), 1)
    hits_differ = int((m.get_stats().num_accumulated_per_pixel.cpu().numpy().reshape(H, W)[crop] != ref["num_accumulated"][crop]).sum())
    report(f"config_b_crop_at_size_vs_oracle[{'help_on' if team_help else 'help_off'}]", pixels=int(crop.sum()), worst=min(lv.values()), pixels_with_other_last_step_hit_count=hits_differ, **lv)
    assert min(lv.values()) >= 50.0, lv
    Ws, Hs = 160, 90
    rt2, o = make_pair(ren, orc, g, cam, Ws, Hs, cfg=dict(jitter_primary_rays=0), fwd=100_000_000, bwd=1_000_000, team_help=team_help)
    with torch.no_grad():
        rt2(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt2)
    lv = {k: round(psnr(out[k], ref[k]), 1) for k in ("output_rgb", "output_final", "output_depth", "output_total_transmittance")}
    report("config_b_lowres_vs_oracle", **lv)
    assert min(lv.values()) > 70, lv


# ------------------------------------------------------------------------------------------------ config 3, init variant
def test_config_c_init_variant_1080p_1M_forward_backward(ren, orc, syn):
    """BASELINE config 3 on the literal dense-INIT cloud (init_opa 0.1, config.py:44): long hit lists (Kc ~ 22 per ray), bounces
    mostly die. Full-size properties: status, counters, gradient linearity in the loss weights, determinism of total_weight."""
    W, H, N = 1920, 1080, 1_000_000
    g = syn.make_scene(N, "init", seed=0)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000)
    m = rt.cuda_module
    camt = cam_obj(ren, cam, tg)
    m.get_config().jitter_primary_rays.fill_(False)
    run_grad(ren, rt, camt)
    c = m.get_counters()
    assert c[11] == 0 and c[0] == W * H
    kc = c[6] / c[0]
    report("config_c_init", rays=list(c[0:3]), composited_per_primary_ray=round(kc, 2), evaluated_per_primary_ray=round(c[3] / c[0], 2))
    assert kc > 15  # the init cloud really has long lists
    g1 = m.get_gaussians().grad_flat.clone()
    assert bool(torch.isfinite(g1).all())
    w = g1[21 * N:]
    assert float(w.sum()) > 0 and float(w.min()) >= 0  # total_weight = sum of compositing weights
    cfg = m.get_config()
    for k in ("loss_weight_diffuse", "loss_weight_specular", "loss_weight_depth", "loss_weight_normal", "loss_weight_f0", "loss_weight_roughness"):
        getattr(cfg, k).mul_(2.0)
    m.get_metadata().total_num_calls.sub_(1)
    run_grad(ren, rt, camt)
    g2 = m.get_gaussians().grad_flat.clone()
    assert float((g2[21 * N:] - w).abs().max()) <= 1e-3 * float(w.abs().max())
    assert float((g2[: 21 * N] - 2 * g1[: 21 * N]).abs().max()) <= 2e-3 * float(g1[: 21 * N].abs().max())


# ------------------------------------------------------------------------------------------------ config 3, gradient check at size
CROP_TILES = [(0, 0), (119, 0), (0, 67), (119, 67), (60, 34), (59, 33), (30, 20), (90, 50), (45, 55), (75, 55), (10, 34), (110, 34), (60, 5), (60, 62),
              (3, 30), (116, 40), (20, 66), (100, 1), (37, 12), (83, 47), (52, 40), (68, 28), (15, 50), (105, 18)]  # 16x16 macro tiles (mx, my): corners,
# the bottom row (half outside the image: 1080 = 67.5 tiles), centre, side walls seen at a grazing angle (image edges), floor / ceiling
CROP_TILES += [(int(x), int(y)) for x, y in zip(np.random.default_rng(11).integers(0, 120, 24), np.random.default_rng(12).integers(0, 68, 24))]
CROP_TILES = sorted(set(CROP_TILES))


# measured on MI355X (profiles/r6/parity_levels.txt; the same in all three modes), asserted with a margin: (clean share of the crop, worst image dB on the clean
# pixels, on all traced pixels, worst gradient error on all traced pixels, pixels with another hit SEQUENCE)
CROP_BARS = {("init", 2): dict(clean=0.99, psnr_clean=85.0, psnr_all=70.0, err_all=1.2e-2, differing=60),       # measured 0.9973, 95.4 dB, 79.5 dB, 7.9e-3, 32
             ("trained", 2): dict(clean=0.95, psnr_clean=75.0, psnr_all=50.0, err_all=1.5e-2, differing=450),   # measured 0.9624, 83.2 dB, 57.9 dB, 9.4e-3, 346
             ("trained", 0): dict(clean=0.995, psnr_clean=105.0, psnr_all=85.0, err_all=3e-3, differing=25)}    # measured 0.9992, 117.6 dB, 96.0 dB, 1.8e-3, 10


@pytest.mark.parametrize("mode", ["help_off", "product_default_help_on", "help_on_eight_ranks_summed"])
@pytest.mark.parametrize("variant,bounces", [("init", 2), ("trained", 2), ("trained", 0)])
def test_config_c_gradient_check_vs_oracle_at_size(ren, orc, syn, variant, bounces, mode):
    """BASELINE config 3 as written: 1M gaussians, 1920x1080, forward + backward, REFERENCE DEFAULTS (jitter on, two bounces,
    training loss weights), "grad check vs ref" AT THE CONFIG'S OWN SIZE. Both sides trace the pixels of 48 macro tiles of the
    full-size frame - same gaussians, same camera, same rays as the whole image - through a PIXEL MASK (oracle: set_pixel_mask; HIP:
    egr_debug_set_pixel_mask, a masked pixel is a pixel outside the image for every kernel of the launch); the HIP launch is a
    whole-image launch of the product (8x8 tasks), its gradients accumulate like the reference's atomicAdds over pixels
    (backward_pass.cu:89-220).

    What fp32 allows at this scale (gaussians of 0.01 units, 20-70 composited hits per primary ray, three steps): whether a ray
    composites one hit more or less - the transmittance threshold, a grazing candidate, a bounce that happens or not - hangs on the
    last bits of exp() and of the bounce direction, and one such hit is up to 1e-2 of a tensor's maximum on a crop of 12k pixels. The
    fp32 oracle disagrees with ITS OWN fp64 evaluation on the hit count of 1-2 % of the pixels. So the check is PIXEL-granular:
      * pass 1 traces the whole crop: images, gradients, per-step hit counts and the ORDERED SEQUENCE of composited gaussians of every pixel and step on
        both sides (as a 64-bit hash: egr_debug_get_hit_sequence_hash from the HIP path's hit arena, Outputs::hit_sequence_hash from the oracle's compositing loop);
      * a pixel is CLEAN when both sides composite the same gaussians in the same order on every step (equal hashes) and its forward outputs agree to 1e-3
        (the total transmittance also sees the candidates behind the last composited hit - quirk Q1); clean pixels must be >= 99 / 95 / 99.5 % of the crop
        (measured 99.7 / 96.2 / 99.9 %; rounds 4-5 compared hit COUNTS and had to exclude every pixel holding two hits within a distance window: 98.0 / 84.1 /
        99.3 %), no more pixels may differ in sequence than between the fp32 and the fp64 oracle (+ 24), and a pixel with another sequence on the PRIMARY step
        must have a REASON the oracle itself reports: its closest yes / no decision (|u|^2 vs 1, T vs the threshold, |normal| vs the bounce threshold:
        Outputs::decision_margin) lies within 1e-3 of flipping, or it holds a near-tie of depths, or the fp32 and the fp64 oracle disagree on it too;
      * pass 2 traces the clean pixels only, on both sides: every step's image >= 75 dB, ALL NINE gradient tensors (total_weight included) < 1e-3 of the
        oracle's max-abs (measured 7.7e-5 / 6.9e-4 / 3.7e-5);
      * on ALL traced pixels: >= 50 dB, gradients < 1.5e-2 (the fp32 oracle is 2.7e-3 ... 1.7e-2 from its own fp64 evaluation).
    The targets are moved off the scene's own wall values (normal, depth, roughness, f0): where an opaque wall renders exactly its
    target, sign(output - target) hangs on the last bit on both sides.

    `mode`: the kernels under test. "help_off" = single-wave workgroups (k_forward_chain<.., 1>, k_backward_chain<1>); "product_default_help_on" = what the
    library ships and bench.py times (teams: k_forward_chain<.., 16>, k_backward_chain<4>, waves without tiles walk other tiles' pairs - in this masked launch
    most waves have none, so help is the rule); "help_on_eight_ranks_summed" = the same crop traced as the EIGHT ranks of an 8-way partition one after the
    other (egr_set_partition(r, 8): every rank's launch is under-filled, help is most of it), images tiled together and gradients summed - the same bars."""
    W, H, N = 1920, 1080, 1_000_000
    par = importlib.import_module(PKG + ".parallel")
    bars = CROP_BARS[(variant, bounces)]
    g = syn.make_scene(N, variant, seed=0)
    cam = syn.default_camera()
    tg = generic_targets(syn, W, H)
    tg["normal"] = tg["normal"] + np.float32([0.11, -0.07, 0.05])
    tg["depth"] = tg["depth"] + np.float32(0.37)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(num_bounces=bounces), fwd=400_000_000, bwd=300_000_000, team_help=(mode != "help_off"))  # reference defaults + the training loss weights
    parts = [(r, 8) for r in range(8)] if mode == "help_on_eight_ranks_summed" else [(0, 1)]
    o64 = orc.Oracle(W, H, double=True)
    o64.set_camera(cam["origin"], cam["c2w"], cam["fov"])
    o64.set_gaussians(g)
    o64.set_config(**o.config)
    o64.update_bvh()
    m = rt.cuda_module
    camt = cam_obj(ren, cam, tg)
    K = 5  # the launch index every launch of this test uses: same jitter, same bounce samples on both sides
    crop = np.zeros((H, W), bool)
    for mx, my in CROP_TILES:
        crop[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] = True

    def hip_on(mask, images=False):
        """One whole-image launch of the product restricted to `mask`: (images of a no-grad launch | None, gradients, per-step hit counts)."""
        m.debug_set_pixel_mask(torch.from_numpy(mask.astype(np.uint8)).cuda())
        try:
            img = None
            if images:
                for name in OUT_KEYS:
                    getattr(m.get_framebuffer(), name).zero_()
                rays = 0
                for r, w in parts:  # (a rank of a partition writes the pixels of its own tiles: the eight launches tile the image together)
                    m.set_partition(r, w)
                    m.get_metadata().total_num_calls.fill_(K - 1)
                    with torch.no_grad():
                        rt(camt)
                    rays += m.get_counters()[0]
                assert rays == int(mask.sum())
                img = hip_outputs(rt)
            rt.zero_grad()
            m.get_gaussians().total_weight.zero_()
            rays, hits, seq = 0, 0, np.zeros((3, H, W), np.uint64)
            for r, w in parts:  # (grad launches ADD to the gradient tensors, like the reference's atomicAdds: eight ranks sum up)
                m.set_partition(r, w)
                m.get_metadata().total_num_calls.fill_(K - 1)
                ren.render(camt, rt)
                assert m.get_counters()[11] == 0
                rays += m.get_counters()[0]
                hits = hits + m.debug_step_hits().numpy()  # (pixels outside the rank's tiles report 0)
                seq = seq | m.debug_hit_sequence_hash().numpy().view(np.uint64)  # ordered composited gaussian ids per pixel and step, hashed (0 outside the rank's tiles)
            assert rays == int(mask.sum())
            return img, hip_grads(rt), hits, seq
        finally:
            m.debug_set_pixel_mask(torch.empty(0, dtype=torch.uint8))
            m.set_partition(0, 1)

    def oracle_on(oo, mask, images=False):
        oo.set_pixel_mask(mask)
        oo.total_num_calls = K - 1
        ref = oo.raytrace(True, targets=tg)
        img = None
        if images:
            oo.total_num_calls = K - 1
            img = oo.raytrace(False)
        oo.set_pixel_mask(None)
        return ref, img

    def image_levels(img_h, img_o, msk):
        lv = {}
        for key in ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_total_transmittance"):
            for s in range(bounces + 1):
                lv[f"{key}[{s}]"] = round(float(psnr(img_h[key][s][msk], img_o[key][s][msk])), 1)
        lv["output_final"] = round(float(psnr(img_h["output_final"][0][msk], img_o["output_final"][0][msk])), 1)
        return lv

    def errors(got, ref_, scale):
        return {k: float(np.abs(got[k] - ref_[k]).max() / np.abs(scale[k]).max()) for k in GRAD_KEYS}

    # ---- pass 1: the whole crop
    img_h, grad_h, hits_h, seq_h = hip_on(crop, images=True)
    ref, img_o = oracle_on(o, crop, images=True)
    ref64, _ = oracle_on(o64, crop)
    levels = image_levels(img_h, img_o, crop)
    err_all = errors(grad_h, ref, ref)
    floor = errors({k: ref64[k] for k in GRAD_KEYS}, ref, ref)
    differing = np.any(hits_h != ref["num_composited_per_step"], axis=0) & crop
    differing_oracles = np.any(ref["num_composited_per_step"] != ref64["num_composited_per_step"], axis=0) & crop
    near_tie = ((ref["num_depth_ties"] > 0) | (ref["num_bounce_near_ties"] > 0)) & crop  # (a REASON for another sequence, no longer a criterion: see `clean`)
    # CLEAN = both sides composite THE SAME GAUSSIANS IN THE SAME ORDER on every step (hashes of the ordered id sequences: egr_debug_get_hit_sequence_hash /
    # Outputs::hit_sequence_hash) - exact, where rounds 4-5 compared hit COUNTS and excluded every pixel with two hits within a distance window
    same_sequence = np.all(seq_h == ref["hit_sequence_hash"], axis=0)
    same_sequence_oracles = np.all(ref["hit_sequence_hash"] == ref64["hit_sequence_hash"], axis=0)
    assert not np.any(same_sequence & crop & np.any(hits_h != ref["num_composited_per_step"], axis=0))  # (equal hashes imply equal counts)
    # (the total transmittance also sees the candidates BEHIND the last composited hit - quirk Q1 -: a pixel whose forward outputs are off by more than
    # 1e-3 anywhere although its sequences agree has met one candidate more or less there; counted, and not clean either)
    off = np.zeros((H, W), bool)
    for key in ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance", "output_total_transmittance"):
        off |= np.any(np.abs(img_h[key] - img_o[key]) > 1e-3, axis=(0, -1))
    off &= crop
    clean = crop & same_sequence & ~off
    # every pixel with another hit count on the PRIMARY step has a reason the oracle reports itself (on a bounce step the ray itself already
    # carries the rounding of the step before it: GGX sampling amplifies the last bits of the accumulated normal)
    thin = ref["decision_margin"] < 1e-3
    differing0 = (seq_h[0] != ref["hit_sequence_hash"][0]) & crop
    unexplained = differing0 & ~thin & ~near_tie & ~differing_oracles & same_sequence_oracles
    # ---- pass 2: the clean pixels only, both sides
    img_hc, grad_hc, hits_hc, seq_hc = hip_on(clean, images=True)
    ref_c, img_oc = oracle_on(o, clean, images=True)
    # (a pixel's rays do not depend on which other pixels are traced: pass 2 composites pass 1's sequences - bit for bit without help; with help the ORDER of
    # exactly tied depths of a bounce ray hangs on timing, DESIGN.md 2 (a): a few pixels may swap two tied hits between the two launches, counted and bounded)
    reordered = int(np.any((seq_hc != ref_c["hit_sequence_hash"]) & clean[None], axis=0).sum())
    assert reordered == 0 if mode == "help_off" else reordered <= 0.005 * int(crop.sum()), reordered
    levels_clean = image_levels(img_hc, img_oc, clean)
    err_clean = errors(grad_hc, ref_c, ref)
    fmt = lambda d: {k: f"{v:.1e}" for k, v in d.items()}
    share = float(clean.sum()) / float(crop.sum())
    report(f"config_c_crop_{variant}_bounces{bounces}[{mode}]", tiles=len(CROP_TILES), pixels=int(crop.sum()), composited=ref["num_composited_per_step"].sum(axis=(1, 2)).tolist(),
           clean_pixels=int(clean.sum()), clean_share=round(share, 4), clean_pixels_that_reordered_an_exact_tie_in_the_second_launch=reordered, pixels_with_another_hit_sequence=int((crop & ~same_sequence).sum()), of_which_on_the_primary_step=int(differing0.sum()),
           pixels_where_fp32_and_fp64_oracle_differ_in_sequence=int((crop & ~same_sequence_oracles).sum()), same_sequence_but_outputs_off_by_1e3=int((crop & same_sequence & off).sum()),
           pixels_with_other_hit_counts=int(differing.sum()), pixels_with_a_near_tie=int(near_tie.sum()),
           differing_with_a_decision_within_1e3=int((differing & thin).sum()), differing_where_fp32_and_fp64_oracle_differ_too=int((differing & differing_oracles).sum()),
           differing_on_the_primary_step=int(differing0.sum()), primary_step_unexplained=int(unexplained.sum()), pixels_with_outputs_off_by_1e3=int(off.sum()), pixels_where_fp32_and_fp64_oracle_differ_in_hit_counts=int(differing_oracles.sum()),
           psnr_min_all_pixels=min(levels.values()), psnr_min_clean_pixels=min(levels_clean.values()), psnr_clean_pixels=levels_clean,
           grad_err_clean_pixels=fmt(err_clean), grad_err_all_pixels=fmt(err_all), fp32_oracle_vs_fp64_oracle_all_pixels=fmt(floor))
    assert share >= bars["clean"], (share, int(differing.sum()), int(near_tie.sum()))
    assert int(unexplained.sum()) <= 3, np.argwhere(unexplained)[:10].tolist()  # (measured 0 / 0-1 / 1: a near-tie between the last composited hit and the first one not reached is not in the oracle's tie count)
    assert min(levels_clean.values()) >= bars["psnr_clean"] and min(levels.values()) >= bars["psnr_all"], (levels_clean, levels)
    assert max(err_clean.values()) < 1e-3, err_clean  # the north-star bar, all nine tensors (round 5 held total_weight of the three-step case to 3e-3: its "clean" pixels still hid swapped hits)
    assert max(err_all.values()) < bars["err_all"], err_all
    other, other_oracles = int((crop & ~same_sequence).sum()), int((crop & ~same_sequence_oracles).sum())
    assert other <= bars["differing"] and other <= other_oracles + 24, (other, other_oracles)  # (absolute: measured + margin; relative: no worse than fp32 against fp64 arithmetic on the oracle's side)


# ------------------------------------------------------------------------------------------------ config scalars
@pytest.mark.parametrize("cfg", [dict(global_scale_factor=2.0), dict(exp_power=2.0), dict(global_scale_factor=0.5, exp_power=4.0, alpha_threshold=0.02)])
def test_non_default_scale_factor_and_exp_power(ren, orc, syn, cfg):
    """global_scale_factor (viewer path, gaussian_viewer.py:287-344) scales the instance transforms but not the backward's
    rot_row = M_row / (s sigma + eps) (backward_pass.cu:178-180 ignores it); exp_power != 3 takes the powf branches."""
    W, H = 80, 48
    g = syn.make_scene(3000, "trained", seed=31)
    cam = syn.default_camera()
    tg = generic_targets(syn, W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, **cfg))
    rt.cuda_module.rebuild_bvh()  # the pair was built before the config scalars were written
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    lv = {k: round(psnr(out[k], ref[k]), 1) for k in ("output_rgb", "output_final", "output_depth", "output_normal", "output_total_transmittance")}
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    refg = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    ge = {k: float(np.abs(gr[k] - refg[k]).max() / (np.abs(refg[k]).max() + 1e-30)) for k in GRAD_KEYS}
    report("cfg_" + "_".join(f"{k}{v}" for k, v in cfg.items()), **lv, worst_grad=f"{max(ge.values()):.1e}")
    assert min(lv.values()) > 60, lv
    assert max(ge.values()) < 1e-3, ge


def test_grad_launch_without_refit_reads_scale_and_rotation_live(ren, orc, syn):
    """backward_pass.cu:68-70 reads scale / rotation from the parameter tensors while M and W come from OptiX's instance snapshot
    (:75-78). The caller always refits before a grad launch (gaussian_raytracer.py:139-140), but raytrace() in grad mode without
    update_bvh() is legal: the gradients must then mix snapshot transforms with live scale / rotation exactly like upstream."""
    W, H = 64, 40
    g = syn.make_scene(2500, "trained", seed=17)
    cam = syn.default_camera()
    tg = generic_targets(syn, W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=1))
    camera = cam_obj(ren, cam, tg)
    run_grad(ren, rt, camera)  # uploads camera + targets, snapshot = g
    g2 = {k: v.copy() for k, v in g.items()}
    rng = np.random.default_rng(3)
    g2["scale"] += rng.uniform(-0.3, 0.3, g2["scale"].shape).astype(np.float32)
    g2["rotation"] += rng.normal(0, 0.3, g2["rotation"].shape).astype(np.float32)
    gs = rt.cuda_module.get_gaussians()
    gs.scale.copy_(torch.tensor(g2["scale"]).cuda())
    gs.rotation.copy_(torch.tensor(g2["rotation"]).cuda())
    rt.zero_grad()
    gs.total_weight.zero_()
    rt.cuda_module.get_metadata().total_num_calls.zero_()
    rt.cuda_module.raytrace()  # grad mode, NO update_bvh: transforms stay the snapshot of g
    torch.cuda.synchronize()
    o.set_gaussians(g2)  # live parameters changed, snapshot kept
    ref = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    ge = {k: float(np.abs(gr[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)) for k in GRAD_KEYS}
    report("live_scale_rotation", **{k: f"{v:.1e}" for k, v in ge.items()})
    assert max(ge.values()) < 1e-3, ge
    o.update_bvh()
    ref_refit = o.raytrace(True, targets=tg)
    assert np.abs(ref_refit["dL_dscale"] - ref["dL_dscale"]).max() > 1e-2 * np.abs(ref["dL_dscale"]).max()  # the two states differ


# ------------------------------------------------------------------------------------------------ robustness
@pytest.mark.parametrize("n,W,H", [(0, 16, 16), (1, 8, 8), (1, 1, 1), (7, 17, 3), (9, 1, 1), (64, 33, 65), (500, 128, 8)])
def test_tiny_models_and_images(ren, orc, syn, n, W, H):
    """N = 0 (the native holder starts with count = 1, core/gaussians.h:31: the caller mirror resizes it) up to a few hundred
    gaussians on 1x1 ... 128x8 images: no exceptions, status 0, finite outputs, images equal to the oracle's."""
    g = {k: v[:n] for k, v in syn.make_scene(max(n, 1), "trained", seed=3).items()}
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=4_000_000, ppll_backward_size=4_000_000)
    m = rt.cuda_module
    assert m.get_gaussians().mean.shape[0] == n and m.check_bvh() == 0
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    out = hip_outputs(rt)
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    c = m.get_counters()
    assert c[11] == 0 and c[0] == W * H
    assert all(np.isfinite(v).all() for k, v in out.items() if k not in ("output_ray_direction", "output_ray_origin"))
    assert bool(torch.isfinite(m.get_gaussians().grad_flat).all())
    if n == 0:
        assert float(np.abs(out["output_final"]).max()) == 0.0 and float(out["output_transmittance"].min()) == 1.0
        return
    o = orc.Oracle(W, H)
    o.set_camera(cam["origin"], cam["c2w"], cam["fov"])
    o.set_config(**syn.TRAIN_LOSS_WEIGHTS)
    o.set_gaussians(g)
    o.update_bvh()
    m.get_metadata().total_num_calls.zero_()
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    assert psnr(hip_outputs(rt)["output_rgb"], o.raytrace(False)["output_rgb"]) > 90


def test_stress_anisotropic_blobs_vs_oracle(ren, orc, syn):
    """Sizes over two orders of magnitude and deep lists (what the bench scene does not have), against the oracle."""
    W, H = 128, 96
    g = syn.random_blob_scene(3000, seed=2, extent=1.0, depth_range=(1.0, 6.0), scale_range=(0.01, 0.4))
    cam = syn.plus_x_camera()
    tg = syn.make_targets(W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0), fwd=50_000_000, bwd=50_000_000)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    ha = rt.cuda_module.get_stats().num_accumulated_per_pixel.cpu().numpy()
    bad, nbad = mismatch_list(ha, ref["num_accumulated"])
    lv = {k: round(psnr(out[k], ref[k]), 1) for k in ("output_rgb", "output_depth", "output_total_transmittance")}
    report("stress_blobs", **lv, acc_mismatch=nbad, first=bad, max_hits=int(ha.max()))
    assert min(lv.values()) > 80 and nbad <= 3
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    refg = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    for k in GRAD_KEYS:
        assert np.abs(gr[k] - refg[k]).max() / (np.abs(refg[k]).max() + 1e-30) < 1e-3, k
    assert rt.cuda_module.get_counters()[11] == 0


def test_stress_3000_candidates_per_ray_overflow_is_flagged(ren, syn):
    """200k large blobs in front of the camera: thousands of candidates per ray. With a small forward budget the run must end with
    the overflow bits set (never out-of-bounds writes: per_pixel_linked_list.h:30-42 has no check upstream), with a large one
    with status 0 and finite gradients."""
    W, H = 320, 180
    g = syn.random_blob_scene(200_000, seed=1, extent=2.0, depth_range=(1.0, 8.0), scale_range=(0.002, 0.3))
    cam = syn.plus_x_camera()
    tg = syn.make_targets(W, H)
    for fwd, bwd, want_ok in ((2_000_000, 2_000_000, False), (400_000_000, 300_000_000, True)):
        rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=fwd, ppll_backward_size=bwd)
        run_grad(ren, rt, cam_obj(ren, cam, tg))
        c = rt.cuda_module.get_counters()
        report("stress_overflow", fwd=fwd, status=c[11], evaluated_per_ray=round(c[3] / c[0], 1), composited_per_ray=round(c[6] / c[0], 1))
        assert (c[11] == 0) == want_ok, c[11]
        assert bool(torch.isfinite(rt.cuda_module.get_gaussians().grad_flat).all())
        del rt
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ host step
def test_render_then_fused_step_counts_raytracer_gradients_once(ren, syn):
    """trainer.FusedTrainStep imports the raytracer gradients inside its kernel, so GaussianRaytracer.__call__ must not add
    them to the model's .grad as well. With a model-side gradient present (a regulariser), the update must equal torch's Adam on
    g = model_grad + rt_grad - not model_grad + 2 rt_grad."""
    tr = importlib.import_module(PKG + ".trainer")
    W, H = 48, 32
    g = syn.make_scene(1500, "trained", seed=5)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000)
    rt.cuda_module.get_config().jitter_primary_rays.fill_(False)
    lrs = dict(xyz=1e-3, normal=1e-3, roughness=1e-3, f0=1e-3, f_dc=1e-3, opacity=1e-2, scaling=1e-3, rotation=1e-3)
    step = tr.FusedTrainStep(pc, rt, lrs)
    assert rt.import_grads is False
    ref_params = [p.clone().requires_grad_(True) for p in pc.parameters()]
    opt = torch.optim.Adam([{"params": [p], "lr": lrs[name]} for p, name in
                            zip(ref_params, ("xyz", "opacity", "scaling", "rotation", "f_dc", "normal", "roughness", "f0"))], eps=1e-15)
    reg = [0.01 * torch.randn_like(p) for p in pc.parameters()]  # a model-side gradient
    for it in range(3):
        for p, r in zip(pc.parameters(), reg):
            p.grad.copy_(r)
        ren.render(cam_obj(ren, cam, tg), rt)
        gs = rt.cuda_module.get_gaussians()
        rt_grads = [gs.mean.grad, gs.opacity.grad, gs.scale.grad, gs.rotation.grad, gs.rgb.grad, gs.normal.grad, gs.roughness.grad, gs.f0.grad]
        for p, mg in zip(pc.parameters(), reg):
            assert torch.equal(p.grad, mg)  # render() left the model gradients alone
        for p, r, rg in zip(ref_params, reg, rt_grads):
            p.grad = r + rg.clone()
        opt.step()
        with torch.no_grad():
            ref_params[4].clamp_(min=0.0), ref_params[6].clamp_(0.0, 1.0), ref_params[7].clamp_(0.0, 1.0)  # train.py:251-254
        step.step()
        torch.cuda.synchronize()
        for k, (p, q) in enumerate(zip(pc.parameters(), ref_params)):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), (it, k)
        assert float(gs.grad_flat[: 21 * 1500].abs().max()) == 0.0  # both zero_grads happened in the kernel
    # optimizer surgery keeps the moments aligned with the parameters
    keep = torch.rand(1500, device="cuda") > 0.3
    step.prune(keep)
    assert step.exp_avg["xyz"].shape[0] == int(keep.sum())
    step.extend(10)
    assert step.exp_avg_sq["rotation"].shape == (int(keep.sum()) + 10, 4) and float(step.exp_avg["xyz"][-10:].abs().max()) == 0.0


def test_resize_keeps_gradient_rows_and_total_weight(ren, syn):
    """core/gaussians.h:64-86 resizes every tensor in place (`resize_`): the first min(old, new) rows of the gradient tensors and of
    total_weight survive (train.py adds far-field points in the middle of a pruning interval without clearing total_weight); the
    growth is zeroed here (uninitialised upstream). `.grad` of every parameter still aliases its dL_d* tensor."""
    W, H = 48, 32
    g = syn.make_scene(1200, "trained", seed=8)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000)
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    m = rt.cuda_module
    before = {k: getattr(m.get_gaussians(), k).clone() for k in GRAD_KEYS}
    assert float(before["total_weight"].abs().max()) > 0 and float(before["dL_dmean"].abs().max()) > 0
    m.resize(1500)
    gs = m.get_gaussians()
    for k in GRAD_KEYS:
        t = getattr(gs, k)
        assert t.shape[0] == 1500 and torch.equal(t[:1200], before[k]) and float(t[1200:].abs().max()) == 0.0, k
    assert gs.mean.grad.data_ptr() == gs.dL_dmean.data_ptr() and gs.rotation.grad.data_ptr() == gs.dL_drotation.data_ptr()
    m.resize(700)
    gs = m.get_gaussians()
    for k in GRAD_KEYS:
        assert torch.equal(getattr(gs, k), before[k][:700]), k
    # the resized tracer still works: new values in, rebuild, grad launch
    g2 = syn.make_scene(700, "trained", seed=9)
    rt.pc.__init__(g2)
    rt.rebuild_bvh()
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    assert m.get_counters()[11] == 0 and bool(torch.isfinite(m.get_gaussians().grad_flat).all())


@pytest.mark.parametrize("cfg", [
    dict(num_bounces=1), dict(num_bounces=5),  # 5 is clamped to MAX_BOUNCES = 2 (shaders.cu:104)
    dict(reflection_invalid_normal_threshold=0.2), dict(reflection_invalid_normal_threshold=0.99),
    dict(backfacing_max_dist=1.0, backfacing_invalid_normal_threshold=0.5), dict(eps_ray_surface_offset=0.05), dict(eps_min_roughness=0.3),
    dict(alpha_threshold=0.05), dict(transmittance_threshold=0.2), dict(transmittance_threshold=0.0), dict(eps_forward_normalization=1e-2),
    dict(eps_scale_grad=1e-3), dict(loss_weight_depth=0.0, loss_weight_specular=0.01), dict(loss_weight_diffuse=0.0, loss_weight_normal=0.0, loss_weight_f0=0.0, loss_weight_roughness=0.0),
], ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
def test_every_config_scalar_is_read_on_the_device(ren, orc, syn, cfg):
    """T3: the 20 scalars of core/config.h:5-26 live in device tensors that Python mutates in place; each non-default value must
    change the HIP result exactly like it changes the oracle's (images of all three steps and all nine gradient tensors)."""
    W, H = 64, 40
    g = syn.make_scene(2500, "trained", seed=41)
    cam = syn.default_camera()
    tg = generic_targets(syn, W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, **cfg))
    rt.cuda_module.rebuild_bvh()  # alpha_threshold enters the instance transforms
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    lv = {k: round(psnr(out[k], ref[k]), 1) for k in ("output_rgb", "output_final", "output_depth", "output_normal", "output_transmittance", "output_total_transmittance")}
    report("cfg_scalar_" + ",".join(f"{k}={v}" for k, v in cfg.items()), min_psnr=min(lv.values()))
    assert min(lv.values()) > 55, lv
    # gradients at the contract's 1e-3; anything above it must come from a handful of LISTED pixels whose bounce rays met a different
    # number of hits in the two implementations, and vanish when those pixels' tiles are taken out on both sides
    err_all, _, _ = grads_vs_oracle_listing_flipped_pixels(ren, rt, o, cam_obj(ren, cam, tg), tg, W, H, "cfg_scalar_grads_" + ",".join(f"{k}={v}" for k, v in cfg.items()))
    for k in ("dL_dnormal", "dL_df0", "dL_droughness"):  # fed by the primary step only (backward_pass.cu:215-219): round-off level on every pixel
        assert err_all.get(k, 0.0) < 1e-5, (k, err_all)


def test_constructor_pose_and_first_launch_contract():
    """The contract the reference's only test of this path exercises (tests/test_gaussian_tracing.py: construct a Raytracer through
    torch.classes, write znear / zfar in place, set_pose with HOST tensors), asserted here with this suite's own values, plus what
    that test leaves out: shape errors surface as RuntimeError (camera.h:63-64), the holder starts with the requested row count and
    `.grad` wired up (gaussians.h:54-61), and the zero-initialised instances the constructor built a tree over (bvh_wrapper.h:17-22)
    can be traced at once."""
    pkg = importlib.import_module(PKG)
    torch.classes.load_library(pkg.GAUSS_TRACER_PATH)
    W, H, N = 1280, 720, 3
    rt = torch.classes.raytracer.Raytracer(W, H, N, 250_000_000, 150_000_000)
    cam = rt.get_camera()
    cam.znear.fill_(0.05)
    cam.zfar.fill_(50.0)
    c2w = torch.tensor([[0.0, 0.0, -1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # host tensors: set_pose copies them to the device
    cam.set_pose(torch.tensor([3.0, -2.0, 0.5]), c2w)
    assert float(cam.znear) == pytest.approx(0.05) and float(cam.zfar) == 50.0 and cam.vertical_fov_radians.is_cuda
    for bad_origin, bad_rot in ((torch.zeros(2), c2w), (torch.zeros(3), torch.eye(2)), (torch.zeros(3, 1), c2w)):
        with pytest.raises(RuntimeError):
            cam.set_pose(bad_origin, bad_rot)
    gs = rt.get_gaussians()
    assert gs.mean.shape == (N, 3) and gs.rotation.shape == (N, 4) and gs.mean.grad.data_ptr() == gs.dL_dmean.data_ptr()
    with torch.no_grad():
        rt.raytrace()
    torch.cuda.synchronize()
    fb = rt.get_framebuffer()
    assert fb.output_rgb.shape == (3, H, W, 3) and fb.output_final.shape == (1, H, W, 3) and fb.target_depth.shape == (H, W, 1)
    assert rt.get_counters()[11] == 0 and bool(torch.isfinite(fb.output_final).all())
    # all-zero raw parameters: opacity sigmoid(0) = 0.5, scale exp(0) = 1, zero quaternion -> NaN transform -> never hit (masked out)
    assert float(fb.output_transmittance[0].min()) == 1.0 and int(rt.get_metadata().total_num_calls) == 1


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_arbitrary_cameras(ren, orc, syn, seed):
    """Random camera rotation, position inside the room, field of view and aspect ratio: primary rays (T4) and everything behind them."""
    rng = np.random.default_rng(seed)
    W, H = [(72, 40), (40, 72), (56, 56)][seed]
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    cam = dict(origin=rng.uniform(-1.0, 1.0, 3).astype(np.float32), c2w=q.astype(np.float32), fov=np.float32(rng.uniform(0.3, 1.4)), znear=np.float32(0.01), zfar=np.float32(999.9))
    g = syn.make_scene(3000, "trained", seed=50 + seed)
    tg = generic_targets(syn, W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=1))
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    lv = {k: round(psnr(out[k], ref[k]), 1) for k in ("output_rgb", "output_final", "output_depth", "output_normal", "output_ray_direction")}
    report(f"camera_{seed}", **lv)
    assert min(lv.values()) > 70, lv
    assert psnr(out["output_rgb"][0], ref["output_rgb"][0]) > 110
    grads_vs_oracle_listing_flipped_pixels(ren, rt, o, cam_obj(ren, cam, tg), tg, W, H, f"camera_{seed}_grads")


# ------------------------------------------------------------------------------------------------ P1 against reference output
def test_caller_mirror_shoots_the_reference_cameras_rays(ren, syn):
    """P1 / T4 pinned by REFERENCE OUTPUT, not by the oracle: the cameras of tests/golden/reference_cameras.npz (scene/cameras.py
    `Camera` run on (R, T, FoVy); rays from `compute_primary_ray_directions` the way prepare_initial_ply.py:58-66 calls it) go
    through renderer.camera_from_RT -> GaussianRaytracer.__call__ (pose flip, set_pose, fov) into the HIP tracer inside the closed
    synthetic room. Step 0 logs the bounce ray (shaders.cu:141-146): next_origin = origin + depth * dir + eps * next_dir, so the
    primary direction the kernel used is normalize(next_origin - eps * next_dir - camera_center) - compared with the reference's."""
    z = np.load(os.path.join(GOLD, "reference_cameras.npz"))
    g = syn.make_scene(20000, "trained", seed=11)
    worst = 0.0
    for i in range(int(z["num_cases"])):
        W, H = (int(x) for x in z[f"c{i}_wh"])
        rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=20_000_000, ppll_backward_size=1_000_000)
        m = rt.cuda_module
        m.get_config().jitter_primary_rays.fill_(False)
        cam = ren.camera_from_RT(z[f"c{i}_R"], z[f"c{i}_T"], float(z[f"c{i}_FoVy"]))
        with torch.no_grad():
            rt(cam)
        fb = m.get_framebuffer()
        no, nd = fb.output_ray_origin[0].double().cpu().numpy(), fb.output_ray_direction[0].double().cpu().numpy()
        depth = fb.output_depth[0, ..., 0].cpu().numpy()
        hit = (np.abs(nd).sum(-1) > 0) & (depth > 0.2)  # pixels whose step 0 went on to sample a bounce
        assert hit.mean() > 0.8, (i, hit.mean())  # closed room: (almost) every primary ray lands on a wall or a sphere
        eps = float(m.get_config().eps_ray_surface_offset)
        v = no - eps * nd - z[f"c{i}_camera_center"].astype(np.float64)
        d = v / np.linalg.norm(v, axis=-1, keepdims=True)
        err = np.abs(d - z[f"c{i}_dirs"])[hit].max()
        worst = max(worst, float(err))
        assert err < 5e-6, (i, err)  # fp32 position round-off over a depth of a few units
        assert m.get_counters()[11] == 0
        del rt
    report("reference_camera_rays", worst_direction_error=f"{worst:.1e}", cameras=int(z["num_cases"]))


def test_camera_and_target_uploads_follow_the_references_copy_semantics(ren, syn):
    """`set_camera` / `set_targets_chw` replace the caller's `copy_` calls (gaussian_raytracer.py:94-137): like those they take tensors of any device, and a
    target of another shape than [C, H, W] raises instead of being reinterpreted (a [H, W, C] image has the same number of elements)."""
    W, H = 40, 24
    g = syn.make_scene(300, "trained", seed=2)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, ppll_forward_size=4_000_000, ppll_backward_size=4_000_000)
    m = rt.cuda_module
    chw = {k: torch.tensor(v).moveaxis(-1, 0).contiguous() for k, v in tg.items()}  # CPU tensors
    m.set_targets_chw(chw["diffuse"], chw["specular"], chw["depth"], chw["normal"], chw["roughness"], chw["f0"])
    fb = m.get_framebuffer()
    assert torch.equal(fb.target_diffuse.cpu().reshape(H, W, 3), torch.tensor(tg["diffuse"]).reshape(H, W, 3))
    with pytest.raises(RuntimeError):
        m.set_targets_chw(torch.tensor(tg["diffuse"]).cuda(), None, None, None, None, None)  # [H, W, 3]: the reference's copy_(moveaxis) raises too
    c = cam_obj(ren, cam)
    m.get_config().jitter_primary_rays.fill_(False)
    imgs = []
    for R, centre in ((c.R.cpu(), c.camera_center.cpu()), (c.R, c.camera_center)):  # CPU tensors, then CUDA tensors: the same pose, the same image
        m.set_camera(R, centre, float(c.FoVy), 0.01, 999.9)
        m.update_bvh(False)
        m.get_metadata().total_num_calls.zero_()  # (the same bounce samples for both launches)
        with torch.no_grad():
            m.raytrace()
        imgs.append(fb.output_final.clone())
    assert torch.equal(imgs[0], imgs[1]) and float(imgs[0].abs().sum()) > 0
