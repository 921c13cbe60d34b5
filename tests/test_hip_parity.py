"""GPU parity tests: the HIP path (through torch.classes.raytracer -> libraytracer.so -> the C ABI of libegr_hip.so)
against the CPU oracle on identical seeded inputs, against the committed golden fixture, and - at the benchmark's
full size - through size-independent properties.

Tolerances (BASELINE.json north_star): PSNR >= 50 dB on images, gradient max-rel-err < 1e-3 relative to the
tensor's max-abs (float atomics are order-nondeterministic). In practice the HIP path sits near fp32 round-off.
"OptiX reference" itself cannot be run anywhere here; see DESIGN.md.
"""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from hip_common import BOTH_HELP_MODES, GOLD, GRAD_KEYS, OUT_KEYS, PKG, cam_obj, hip_grads, hip_outputs, make_pair, mismatch_list, psnr, ren, report, run_grad  # noqa: F401


# ------------------------------------------------------------------------------------------------ K1/K2
def test_instance_records_match_oracle(ren, orc, syn):
    g = syn.random_blob_scene(500, seed=5)
    g["opacity"][::7] = -8.0  # sigmoid < alpha_threshold -> masked out (bvh_wrapper.cu:55)
    rt, o = make_pair(ren, orc, g, syn.plus_x_camera(), 32, 32)
    M, Wm, A = [t.numpy() for t in rt.cuda_module.debug_instances()]
    Mo, Wo, Ao, vis = o.instances()
    v = vis.astype(bool)
    assert v.sum() < len(v) and (~v).sum() == len(g["opacity"][::7])
    np.testing.assert_allclose(M[v], Mo[v], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(Wm[v], Wo[v], rtol=2e-5, atol=2e-5)
    # the tree bounds each gaussian's ELLIPSOID (half-extent = |row of M|_2), padded, never tighter; it lies inside the
    # cube box the oracle reports (half-extent = |row of M|_1)
    ext = np.sqrt((Mo[:, :, :3] ** 2).sum(-1))
    ctr = Mo[:, :, 3]
    assert np.all(A[v, :3] <= (ctr - ext)[v] + 1e-6) and np.all(A[v, 3:] >= (ctr + ext)[v] - 1e-6)
    assert np.all(A[v, 3:] - A[v, :3] <= 2 * ext[v] * 1.001 + 1e-4)
    assert np.all(A[v, :3] >= Ao[v, :3] - 1e-4) and np.all(A[v, 3:] <= Ao[v, 3:] + 1e-4)
    assert np.all(A[~v, 0] > A[~v, 3])  # invisible -> empty box


def test_instance_records_match_the_reference_build_scaling_rotation(ren, orc, syn):
    """K1 on the GPU against reference OUTPUT (tests/golden/reference_scaling_rotation.npz, produced by running the reference's
    build_scaling_rotation, utils/general_utils.py:79-113): k_instances' M rows = sigma * g * R(q) . diag(exp(scale))."""
    from test_oracle_known_answers import _scaling_rotation_scene

    g, z, sigma = _scaling_rotation_scene()
    rt, o = make_pair(ren, orc, g, syn.plus_x_camera(), 16, 16, cfg=dict(global_scale_factor=1.7))
    rt.cuda_module.update_bvh()  # (the config scalar was written after the constructor's build)
    M, Wm, _ = [t.numpy().astype(np.float64) for t in rt.cuda_module.debug_instances()]
    L = z["L"].astype(np.float64) * sigma * 1.7
    assert np.abs(M[:, :, :3] - L).max() <= 4e-6 * np.abs(L).max(), np.abs(M[:, :, :3] - L).max()
    assert np.abs(np.einsum("nij,njk->nik", Wm[:, :, :3], L) - np.eye(3)).max() < 2e-5


def test_bvh_consistent_after_rebuild_and_refit(ren, orc, syn):
    for n in (1, 2, 3, 17, 1000, 20000):
        g = syn.make_scene(n, "trained", seed=n) if n >= 100 else syn.random_blob_scene(n, seed=n)
        rt, _ = make_pair(ren, orc, g, syn.default_camera(), 16, 16)
        m = rt.cuda_module
        assert m.check_bvh() == 0, m.last_error()
        gs = m.get_gaussians()
        gs.mean.add_(0.05 * torch.randn_like(gs.mean))  # move everything, refit only
        gs.scale.add_(0.1)
        m.update_bvh()
        assert m.check_bvh() == 0, m.last_error()


def test_duplicate_positions_build_a_valid_tree(ren, orc, syn):
    g = syn.random_blob_scene(256, seed=1)
    g["mean"][:] = g["mean"][0]  # identical Morton codes: Karras falls back to index bits
    rt, o = make_pair(ren, orc, g, syn.plus_x_camera(), 16, 16, cfg=dict(jitter_primary_rays=0, num_bounces=0))
    assert rt.cuda_module.check_bvh() == 0
    with torch.no_grad():
        rt(cam_obj(ren, syn.plus_x_camera()))
    ref = o.raytrace(False)
    assert psnr(hip_outputs(rt)["output_rgb"], ref["output_rgb"]) > 50  # 256 concentric Gaussians: near-tied depths


def test_q3_tie_drop_at_16_hit_batch_boundaries_matches_oracle(ren, orc, syn):
    """Quirk Q3 (forward_pass.cu:62): the reference selects hits 16 at a time with a strict `>` against the last composited
    distance, so hits that tie with the 16th, 32nd, ... composited hit of their ray are DROPPED (how many is deterministic,
    which ones is its list order). 20 bit-identical copies of every gaussian: all depths tie 20-fold, every ray composites the
    first 16 copies of each surfel it meets and loses 4 - an order-independent known answer, and the oracle restates the rule."""
    W, H = 48, 32
    g = syn.make_scene(600, "init", seed=13)  # opacity 0.1: alpha <= 0.1, 16 hits leave T >= 0.18 > the threshold
    g20 = {k: np.concatenate([v] * 20, 0) for k, v in g.items()}
    cam = syn.default_camera()
    # (threshold 0: no early stop in the middle of a batch, so every ray composites whole batches of 16 tied copies)
    rt, o = make_pair(ren, orc, g20, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0, transmittance_threshold=0.0))
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    ha = rt.cuda_module.get_stats().num_accumulated_per_pixel.cpu().numpy()
    bad, nbad = mismatch_list(ha, ref["num_accumulated"])
    report("q3_ties", mismatching_pixels=nbad, first=bad, max_hits=int(ha.max()),
           psnr_rgb=round(psnr(out["output_rgb"][0], ref["output_rgb"][0]), 1))
    assert nbad == 0, bad
    assert int(ha.max()) >= 32 and np.all(ha % 16 == 0)  # whole batches of 16 ties, more than one batch somewhere
    T, Tt = out["output_transmittance"][0], out["output_total_transmittance"][0]
    assert bool((Tt < 0.9 * T).any())  # the dropped copies still count in T_total (shaders.cu:69-71)
    for k in ("output_rgb", "output_transmittance", "output_total_transmittance", "output_depth", "output_final"):
        assert psnr(out[k], ref[k]) > 100, k


def test_exact_depth_ties_inside_a_batch_are_all_composited(ren, orc, syn):
    """Two bit-identical copies of every gaussian: ties come in pairs, so no pair straddles a 16-hit batch boundary (hits are
    taken in pairs from an even position) and nothing is dropped; both copies are composited exactly once (a copy taken
    twice would show up as T < T_total). Rays carry several batches, so the 8-way sorted insertion and the batch hand-over
    both meet equal keys."""
    W, H = 64, 48
    g = syn.make_scene(1500, "init", seed=13)  # opacity 0.1: long hit lists
    g2 = {k: np.concatenate([v, v], 0) for k, v in g.items()}
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g2, cam, W, H, cfg=dict(jitter_primary_rays=0))
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    assert np.all(out["output_total_transmittance"] <= out["output_transmittance"] + 4e-6)
    st = rt.cuda_module.get_stats()
    assert int(st.num_accumulated_per_pixel.max().item()) > 16  # more than one reference batch somewhere
    for k in ("output_rgb", "output_transmittance", "output_total_transmittance", "output_final"):
        assert psnr(out[k], ref[k]) > 70, k
    ha = st.num_accumulated_per_pixel.cpu().numpy()
    bad, nbad = mismatch_list(ha, ref["num_accumulated"])  # (last executed step: bounce rays differ by ulps between the two)
    report("pair_ties", mismatching_pixels=nbad, of=W * H, first=bad)
    assert nbad <= 6, bad


# ------------------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("variant", ["trained", "init"])
def test_forward_strict_parity_primary(ren, orc, syn, variant):
    """jitter off, num_bounces 0: the strict-parity configuration of SURVEY.md 8d. Integer outputs are compared exactly; the
    pixels that differ are listed (two implementations round a distance differently by an ulp and a grazing candidate flips
    across a clip test - expf / the fma contraction differ between the CPU and the GPU), not tolerated by fraction."""
    W, H = 96, 64
    g = syn.make_scene(4000, variant, seed=11)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0))
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    st = rt.cuda_module.get_stats()
    ht, ha = st.num_traversed_per_pixel.cpu().numpy(), st.num_accumulated_per_pixel.cpu().numpy()
    bad_acc, n_acc = mismatch_list(ha, ref["num_accumulated"])
    worst = {}
    for k in OUT_KEYS:
        # Two hits whose depths differ by an ulp may composite in the other order: same hit count and transmittance, slightly
        # different colour. Such pixels must agree in everything order-independent; all others agree to 2e-4 absolute.
        err = np.abs(out[k] - ref[k]).reshape(3, H * W, -1).max(-1)[0]
        swapped = err >= 2e-4
        worst[k] = (int(swapped.sum()), round(psnr(out[k], ref[k]), 1))
        assert swapped.sum() <= 2, (k, int(swapped.sum()))
        flip = ha.reshape(-1)[swapped] != ref["num_accumulated"].reshape(-1)[swapped]
        assert flip.sum() <= n_acc, k
        assert err.max() < 1e-2, k
        assert psnr(out[k], ref[k]) > 90, k
    assert np.abs(out["output_transmittance"] - ref["output_transmittance"]).max() < 2e-6
    report("strict_primary_" + variant, acc_mismatch=n_acc, acc_first=bad_acc, swapped_and_psnr=worst)
    assert n_acc <= 1, bad_acc  # measured: 0 on both variants
    # default launches count the candidates whose response point lies INSIDE the gaussian's ellipsoid on the ray's segment (a property of
    # ray and gaussian, whatever the walk; include/egr_raytracer.h: egr_set_exact_stats) - a subset of the reference's intersection
    # invocations (cube overlaps), pixel by pixel; that count itself is compared in test_exact_stats_mode_counts_reference_invocations
    ratio = float(ht.sum()) / float(ref["num_traversed"].sum())
    report("default_statistic_" + variant, inside_ellipsoid_over_reference_invocations=round(ratio, 3))
    assert 0.3 < ratio <= 1.0 and (ht > ref["num_traversed"]).sum() <= 2, (ratio, int((ht > ref["num_traversed"]).sum()))
    assert (ht >= ha).all()  # every composited hit is one of them
    seeds = rt.cuda_module.get_metadata().random_seeds.cpu().numpy().astype(np.uint32).reshape(H, W)
    assert np.array_equal(seeds, ref["random_seeds"].reshape(H, W))
    c = rt.cuda_module.get_counters()
    assert c[0] == W * H and c[1] == 0 and c[3] == int(ht.sum()) and c[11] == 0


@pytest.mark.parametrize("variant,bounces", [("trained", 0), ("trained", 2), ("init", 2)])
def test_exact_stats_mode_counts_reference_invocations(ren, orc, syn, variant, bounces):
    """T7: with set_exact_stats(True) the tree bounds the instance cubes and num_traversed_per_pixel is the reference's
    number - invocations of the intersection program, all steps (shaders.cu:33, forward_pass.cu:46) - equal to the oracle's
    pixel by pixel (listed exceptions: a cube that a ray grazes within an ulp). Images do not depend on the mode."""
    W, H = 96, 64
    g = syn.make_scene(4000, variant, seed=11)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=bounces))
    m = rt.cuda_module
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    base = hip_outputs(rt)
    sub = m.get_stats().num_traversed_per_pixel.cpu().numpy().copy()
    m.set_exact_stats(True)
    with pytest.raises(RuntimeError):  # the boxes are still ellipsoid boxes
        with torch.no_grad():
            rt(cam_obj(ren, cam))
    m.get_metadata().total_num_calls.zero_()
    with torch.no_grad():
        rt(cam_obj(ren, cam), force_update_bvh=True)
    assert m.check_bvh() == 0, m.last_error()
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    ht = m.get_stats().num_traversed_per_pixel.cpu().numpy()
    bad, nbad = mismatch_list(ht, ref["num_traversed"])
    c = m.get_counters()
    report(f"exact_stats_{variant}_b{bounces}", mismatching_pixels=nbad, of=W * H, first=bad, hc_exact=int(ht.sum()), hc_oracle=int(ref["num_traversed"].sum()),
           evaluated_default=int(sub.sum()))
    if bounces == 0:
        assert nbad <= 2, bad  # measured 0
        assert abs(int(ht.sum()) - int(ref["num_traversed"].sum())) <= 2
    else:  # bounce rays differ by ulps between the implementations: the counts of those steps agree statistically
        assert nbad <= 0.02 * W * H, (nbad, bad)
        assert abs(int(ht.sum()) - int(ref["num_traversed"].sum())) <= 2e-3 * int(ref["num_traversed"].sum())
    assert c[3] + c[4] + c[5] == int(ht.sum()) and c[11] == 0
    for k in ("output_rgb", "output_depth", "output_normal", "output_transmittance", "output_total_transmittance", "output_final"):
        if k != "output_final":
            assert psnr(out[k][0], base[k][0]) > 110, k  # step 0: same accepted set, same order (T_total: another product order)
        assert psnr(out[k], ref[k]) > (90 if bounces == 0 else 50), k
    # gradients come out the same in both modes
    tg = syn.make_targets(W, H)
    m.get_metadata().total_num_calls.zero_()
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    ge = hip_grads(rt)
    m.set_exact_stats(False)
    m.get_metadata().total_num_calls.zero_()
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    gd = hip_grads(rt)
    for k in GRAD_KEYS:
        assert np.abs(ge[k] - gd[k]).max() / (np.abs(gd[k]).max() + 1e-30) < 1e-4, k


@BOTH_HELP_MODES
def test_forward_parity_with_bounces_and_jitter(ren, orc, syn, team_help):
    W, H = 96, 64
    g = syn.make_scene(4000, "trained", seed=12)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, team_help=team_help)  # reference defaults: jitter on, 2 bounces
    for call in range(2):  # a different jitter pattern per call (seed = tea4(pixel, total_num_calls))
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        ref = o.raytrace(False)
        out = hip_outputs(rt)
        assert int(rt.cuda_module.get_metadata().total_num_calls.item()) == o.total_num_calls
        levels = {}
        for k in ("output_rgb", "output_final", "output_normal", "output_depth"):
            for s in range(out[k].shape[0]):
                levels[f"{k}[{s}]"] = round(psnr(out[k][s], ref[k][s]), 1)
        report(f"bounces_jitter_call{call}", **levels)
        # the north-star bar is 50 dB; the asserted levels are what is measured (step 0 differs by round-off; a bounce ray that
        # differs by an ulp may meet another grazing candidate, which is what the lower levels of steps 1 and 2 are)
        bars = {"output_rgb[0]": 120, "output_depth[0]": 115, "output_normal[0]": 80, "output_rgb[1]": 95, "output_depth[1]": 95, "output_normal[1]": 90,
                "output_rgb[2]": 70, "output_depth[2]": 60, "output_normal[2]": 50, "output_final[0]": 70}
        for k, v in levels.items():
            assert v > bars[k], (k, v, call)
        assert (ref["effective_steps"] > 1).mean() > 0.5  # the bounce steps were really exercised


@BOTH_HELP_MODES
def test_golden_fixture(ren, orc, syn, team_help):
    z = np.load(os.path.join(GOLD, "scene_2k_64.npz"))
    W, H = int(z["W"]), int(z["H"])
    g = {k[2:]: z[k] for k in z.files if k.startswith("g_")}
    cam = {k[4:]: z[k] for k in z.files if k.startswith("cam_")}
    tg = {k[3:]: z[k] for k in z.files if k.startswith("tg_")}
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=8_000_000, ppll_backward_size=8_000_000, team_help=team_help)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    out = hip_outputs(rt)
    for k in ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance",
              "output_total_transmittance", "output_final"):
        assert psnr(out[k], z["ref_" + k]) > 50.0, k
    st = rt.cuda_module.get_stats()
    ht = st.num_traversed_per_pixel.cpu().numpy()
    assert 0.3 * z["ref_num_traversed"].sum() < ht.sum() <= z["ref_num_traversed"].sum()  # candidates inside their ellipsoid (default mode): a subset of the reference's invocations
    assert (st.num_accumulated_per_pixel.cpu().numpy() != z["ref_num_accumulated"]).mean() < 5e-3
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    gr = hip_grads(rt)
    for k in GRAD_KEYS:
        ref = z["ref_" + k]
        assert np.abs(gr[k] - ref).max() / (np.abs(ref).max() + 1e-30) < 1e-3, k


# ------------------------------------------------------------------------------------------------ backward
@BOTH_HELP_MODES
@pytest.mark.parametrize("bounces", [0, 2])
def test_backward_parity(ren, orc, syn, bounces, team_help):
    W, H = 80, 48
    g = syn.make_scene(3000, "trained", seed=21)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=bounces), team_help=team_help)
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    ref = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    for k in GRAD_KEYS:
        scale = np.abs(ref[k]).max()
        assert scale > 0, k
        assert np.abs(gr[k] - ref[k]).max() / scale < 1e-3, (k, np.abs(gr[k] - ref[k]).max() / scale)
    assert rt.cuda_module.get_counters()[11] == 0


@BOTH_HELP_MODES
def test_backward_parity_long_bounce_chains(ren, orc, syn, team_help):
    """Bounce rays through a translucent cloud composite 20+ hits: their backward chains span several arena blocks per bounce step, which
    the two-pass bounce backward (suffix sums per ray, geometry per hit; backward_task.inc) walks in chunks of four rows. Surfaces are made
    specular enough for the bounces to happen at all (a dense-init cloud's accumulated normal is too short for most rays)."""
    from hip_common import generic_targets, grads_vs_oracle_listing_flipped_pixels

    W, H = 64, 48
    g = syn.make_scene(4000, "init", seed=13)
    g["opacity"] = np.full_like(g["opacity"], np.log(0.35 / 0.65)).astype(np.float32)  # sigmoid^-1(0.35): long lists AND a usable normal
    cam = syn.default_camera()
    tg = generic_targets(syn, W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=2), team_help=team_help)
    grads_vs_oracle_listing_flipped_pixels(ren, rt, o, cam_obj(ren, cam, tg), tg, W, H, "long_bounce_chains_grads")
    hits = rt.cuda_module.debug_step_hits().numpy()  # [3,H,W] of the last grad launch
    report("long_bounce_chains", max_hits_per_step=[int(hits[s].max()) for s in range(3)], bounce_rays=int((hits[1] > 0).sum()))
    assert int(hits[1].max()) > 16 and int((hits[1] > 8).sum()) > 50  # chains of three and more blocks exist, many of two
    assert rt.cuda_module.get_counters()[11] == 0


def test_grad_mode_writes_no_images_and_accumulates_grads(ren, orc, syn):
    """Quirk Q7 (shaders.cu:155-169): outputs are only written when grads are disabled; grads add up across calls."""
    W, H = 32, 32
    g = syn.make_scene(1500, "trained", seed=3)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
    fb = rt.cuda_module.get_framebuffer()
    fb.output_rgb.fill_(123.0)
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    assert float(fb.output_rgb.min()) == 123.0
    g1 = hip_grads(rt)["dL_dmean"].copy()
    pg1 = rt.pc._xyz.grad.clone()
    rt.cuda_module.get_metadata().total_num_calls.sub_(1)  # same RNG stream (bounce sampling is seeded by the call counter)
    ren.render(cam_obj(ren, cam, tg), rt)  # no zero_grad: native grads accumulate (atomicAdd onto existing, backward_pass.cu:210)
    torch.cuda.synchronize()
    np.testing.assert_allclose(hip_grads(rt)["dL_dmean"], 2 * g1, rtol=1e-3, atol=1e-4 * np.abs(g1).max())
    # python-side import is add_: 1x after the first call, + 2x (accumulated native buffer) after the second
    np.testing.assert_allclose(rt.pc._xyz.grad.cpu().numpy(), 3 * pg1.cpu().numpy(), rtol=1e-3, atol=1e-4 * float(pg1.abs().max()))


# ------------------------------------------------------------------------------------------------ semantics
@pytest.mark.parametrize("bounces", [0, 2])
def test_update_bvh_snapshot_semantics(ren, orc, syn, bounces):
    """No-grad renders do not refresh the transforms (gaussian_raytracer.py:139): traversal uses the snapshot while
    alpha / sigma / appearance read the live tensors. The 0.3 shift also pushes boxes out of the build
    frame (5 % head-room), so the refit part runs the walks' out-of-frame (sentinel) decode - packet and group walk."""
    W, H = 48, 32
    g = syn.make_scene(2000, "trained", seed=9)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=bounces))
    g2 = {k: v.copy() for k, v in g.items()}
    g2["mean"] += 0.3  # moved geometry ...
    g2["rgb"] = 1.0 - g2["rgb"]  # ... and recoloured
    rt.pc._xyz.copy_(torch.tensor(g2["mean"]).cuda())
    rt.pc._diffuse.copy_(torch.tensor(g2["rgb"]).cuda())
    with torch.no_grad():
        rt(cam_obj(ren, cam))  # exports the new values, but does NOT call update_bvh
    o.set_gaussians(g2)  # oracle: live params changed, snapshot kept
    ref = o.raytrace(False)
    assert psnr(hip_outputs(rt)["output_rgb"][0], ref["output_rgb"][0]) > 60
    with torch.no_grad():
        rt(cam_obj(ren, cam), force_update_bvh=True)
    o.update_bvh()
    ref2 = o.raytrace(False)
    assert psnr(hip_outputs(rt)["output_rgb"][0], ref2["output_rgb"][0]) > 60
    assert psnr(hip_outputs(rt)["output_final"], ref2["output_final"]) > 55  # bounce steps included
    assert psnr(ref["output_rgb"][0], ref2["output_rgb"][0]) < 40  # the two states really differ


def test_update_bvh_fuse_live_is_the_same_launch_and_later_launches_read_live_values(ren, orc, syn):
    """update_bvh(fuse_live=True) (egr_update_bvh_ex, what GaussianRaytracer.__call__ uses) writes the live records in the transform pass and the
    NEXT raytrace skips its own pass: outputs and gradients must be bit-identical to update_bvh() + raytrace(); the launch after that reads
    the live tensors again (recolouring without update_bvh shows), and a rebuild between the fused update and the launch drops the flag
    (the records move)."""
    W, H = 64, 48
    g = syn.make_scene(3000, "trained", seed=4)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=2))
    m = rt.cuda_module
    tg = syn.make_targets(W, H)
    with torch.no_grad():
        rt(cam_obj(ren, cam, tg))  # camera, targets, parameter export

    def launch(fused, grads, rebuild=False):
        m.get_metadata().total_num_calls.zero_()
        rt.zero_grad()
        m.get_gaussians().total_weight.zero_()
        m.update_bvh(True) if fused else m.update_bvh()
        if rebuild:
            m.rebuild_bvh()
        with torch.set_grad_enabled(grads):
            m.raytrace()
        torch.cuda.synchronize()
        return hip_grads(rt) if grads else hip_outputs(rt)

    a, b = launch(False, False), launch(True, False)
    for k in OUT_KEYS:
        assert np.array_equal(a[k], b[k]), k
    # gradients: same records, same arithmetic; the atomics' order is the only freedom
    ga, gb = launch(False, True), launch(True, True)
    for k in GRAD_KEYS:
        assert np.abs(ga[k] - gb[k]).max() <= 1e-5 * max(float(np.abs(ga[k]).max()), 1e-30), k
    # the launch AFTER the fused pair reads the live tensors again
    launch(True, False)
    m.get_gaussians().rgb.copy_(1.0 - m.get_gaussians().rgb)
    m.get_metadata().total_num_calls.zero_()
    with torch.no_grad():
        m.raytrace()
    recoloured = hip_outputs(rt)
    assert psnr(recoloured["output_rgb"][0], a["output_rgb"][0]) < 40
    g2 = {k: v.copy() for k, v in g.items()}
    g2["rgb"] = (1.0 - g["rgb"]).astype(np.float32)
    o.set_gaussians(g2)
    o.total_num_calls = 0
    ref = o.raytrace(False)
    assert psnr(recoloured["output_rgb"][0], ref["output_rgb"][0]) > 60
    # fused update, then a rebuild (records move), then the launch: same picture as the plain sequence
    c = launch(True, False, rebuild=True)
    for k in ("output_rgb", "output_final", "output_depth"):
        assert psnr(c[k], recoloured[k]) > 100, k


def test_accumulate_samples(ren, orc, syn):
    W, H = 40, 24
    g = syn.make_scene(1500, "trained", seed=4)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(accumulate_samples=1))
    m = rt.cuda_module
    m.reset_accumulators()
    for k in range(3):
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        ref = o.raytrace(False)
        out = hip_outputs(rt)
        assert int(m.get_framebuffer().accumulated_sample_count.item()) == k + 1
        assert psnr(out["output_rgb"], ref["output_rgb"]) > 55, k
        assert psnr(out["output_final"], ref["output_final"]) > 55, k
        np.testing.assert_allclose(out["output_final"][0], out["output_rgb"].sum(0), atol=1e-5)


def test_near_plane_cut_and_far_plane(ren, orc, syn):
    """render.py:36 uses znear = 1.0: candidates in front of it still enter T_total (quirk Q1)."""
    W, H = 48, 32
    g = syn.make_scene(3000, "trained", seed=6)
    cam = dict(syn.default_camera())
    cam["znear"], cam["zfar"] = np.float32(1.5), np.float32(3.0)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=1))
    with torch.no_grad():
        rt(cam_obj(ren, cam), znear=1.5, zfar=3.0)
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    for k in ("output_rgb", "output_total_transmittance", "output_transmittance", "output_depth"):
        assert psnr(out[k], ref[k]) > 55, k


def test_odd_image_sizes_and_resize(ren, orc, syn):
    g = syn.make_scene(1200, "trained", seed=8)
    cam = syn.default_camera()
    for W, H in ((33, 17), (8, 8), (100, 7)):
        rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        assert psnr(hip_outputs(rt)["output_final"], o.raytrace(False)["output_final"]) > 55, (W, H)
    # topology change: grow the model, rebuild (gaussian_raytracer.py:33-38)
    g2 = syn.make_scene(2500, "trained", seed=8)
    rt.pc.__init__(g2)
    rt.rebuild_bvh()
    gs = rt.cuda_module.get_gaussians()
    assert gs.mean.shape[0] == 2500 and gs.mean.grad.data_ptr() == gs.dL_dmean.data_ptr()
    assert float(gs.grad_flat.abs().max()) == 0.0  # grown gradient memory is zeroed
    o.set_gaussians(g2)
    o.update_bvh()
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    assert psnr(hip_outputs(rt)["output_final"], o.raytrace(False)["output_final"]) > 55


def test_nan_rays_match_oracle_positions(ren, orc, syn):
    """Normals exactly (0,0,-1) make sample_cook_torrance return a NaN direction upstream (ggx_brdf.h:163); fmaxf then
    zeroes the throughput, the NaN ray hits nothing, and the pixel stays finite. Same NaN positions and values here."""
    W, H = 32, 32
    g = syn.make_scene(1500, "trained", seed=2)
    g["normal"][:] = np.array([0, 0, -1.0], np.float32)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    assert np.isnan(ref["output_ray_direction"][0]).any() and not np.isnan(ref["output_final"]).any()
    for k in ("output_ray_direction", "output_ray_origin", "output_final", "output_rgb"):
        assert np.array_equal(np.isnan(out[k]), np.isnan(ref[k])), k
    assert psnr(out["output_final"], ref["output_final"]) > 60


def test_capacity_overflow_is_flagged_not_silent(ren, orc, syn):
    W, H = 64, 64
    g = syn.make_scene(20000, "init", seed=1)
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=1000, ppll_backward_size=1000)  # far too small
    cam = syn.default_camera()
    run_grad(ren, rt, cam_obj(ren, cam, syn.make_targets(W, H)))
    status = rt.cuda_module.get_counters()[11]
    assert status & 2, "hit-arena overflow must be reported"  # upstream writes out of bounds here (per_pixel_linked_list.h:30-42)


@BOTH_HELP_MODES
def test_candidate_lists_longer_than_capacity_continue_in_extension_blocks(ren, orc, syn, team_help):
    """ppll_forward_size so small that a ray's own run holds 64 candidates: longer lists spill into extension blocks (the
    reference's pool is global: a single ray may use any number of entries) and the images / gradients still match."""
    W, H = 96, 64
    g = syn.make_scene(6000, "init", seed=3)  # opacity 0.1: long candidate lists
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0), fwd=1000, bwd=50_000_000, team_help=team_help)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    c = rt.cuda_module.get_counters()
    st = rt.cuda_module.get_stats()
    assert c[11] == 0, "no overflow: the lists continue in extension blocks"
    assert int(st.num_traversed_per_pixel.max().item()) > 150  # far more candidates on some rays than one 64-entry run holds
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    for k in ("output_rgb", "output_final", "output_total_transmittance"):
        assert psnr(out[k], ref[k]) > 80, k
    # gradients: identical (up to float-atomic order) to a run whose lists fit their own runs
    big, _ = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0), fwd=50_000_000, bwd=50_000_000, team_help=team_help)
    for r in (rt, big):
        r.cuda_module.get_metadata().total_num_calls.zero_()
        run_grad(ren, r, cam_obj(ren, cam, tg))
        assert r.cuda_module.get_counters()[11] == 0
    hs, hb = hip_grads(rt), hip_grads(big)
    for k in GRAD_KEYS:
        assert np.abs(hs[k] - hb[k]).max() / (np.abs(hb[k]).max() + 1e-30) < 1e-5, k


@pytest.mark.parametrize("rays_per_task", [64, 32, 16])
def test_tile_partition_sums_to_full_image(ren, orc, syn, monkeypatch, rays_per_task):
    """Multi-GPU split on one device: rank r of 2 traces its tiles only; images tile together, gradients add up - with 8x8-pixel
    tasks and with the 8x4 / 4x4 tasks an under-filled rank switches to (env EGR_RAYS_PER_TASK pins the shape for every tracer of
    this test: the order of a ray's candidate list, hence of exactly tied hits, depends on it)."""
    monkeypatch.setenv("EGR_RAYS_PER_TASK", str(rays_per_task))
    W, H = 80, 48
    g = syn.make_scene(3000, "trained", seed=5)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    full, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
    with torch.no_grad():
        full(cam_obj(ren, cam))
    img_full = hip_outputs(full)["output_final"]
    assert psnr(img_full, o.raytrace(False)["output_final"]) > 55  # (this task shape against the oracle, whole image)
    run_grad(ren, full, cam_obj(ren, cam, tg))
    gfull = hip_grads(full)
    refg = o.raytrace(True, targets=tg)
    for k in GRAD_KEYS:
        assert np.abs(gfull[k] - refg[k]).max() / (np.abs(refg[k]).max() + 1e-30) < 1e-3, (rays_per_task, k)
    parts, gparts, counters = [], [], []
    for r in range(2):
        rt, orr = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
        rt.cuda_module.set_partition(r, 2)
        rt.cuda_module.get_framebuffer().output_final.zero_()
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        parts.append(hip_outputs(rt)["output_final"])
        counters.append(rt.cuda_module.get_counters()[0])
        run_grad(ren, rt, cam_obj(ren, cam, tg))
        gparts.append(hip_grads(rt))
        if r == 0:  # oracle's partition agrees with the HIP partition pixel for pixel
            orr.set_partition(0, 2)
            ref0 = orr.raytrace(False)["output_final"]
            assert psnr(parts[0], ref0) > 55
    assert counters[0] + counters[1] == W * H
    assert np.all((parts[0] == 0) | (parts[1] == 0))
    np.testing.assert_allclose(parts[0] + parts[1], img_full, atol=1e-6)
    for k in GRAD_KEYS:
        s = gparts[0][k] + gparts[1][k]
        assert np.abs(s - gfull[k]).max() / (np.abs(gfull[k]).max() + 1e-30) < 1e-3, k


@pytest.mark.parametrize("size", [(80, 48), (37, 29)])
def test_targets_upload_equals_the_references_per_buffer_copies(ren, orc, syn, size):
    """`set_targets_chw` (one launch) against what the reference's caller does per buffer (`renderer.py:118-147`:
    `framebuffer.target_x.copy_(image.moveaxis(0, -1))`, a missing image zeroes the buffer): whole image (16-B path when the pixel
    count allows, scalar path for 37 x 29) and a rank of two (own tiles only, the other pixels keep what they held)."""
    W, H = size
    g = syn.make_scene(500, "trained", seed=3)
    cam = syn.default_camera()
    gen = torch.Generator(device="cpu").manual_seed(11)
    chans = dict(diffuse=3, specular=3, depth=1, normal=3, roughness=1, f0=3)
    imgs = {k: torch.rand(c, H, W, generator=gen).cuda() for k, c in chans.items()}
    order = ["diffuse", "specular", "depth", "normal", "roughness", "f0"]
    for world, missing in ((1, ()), (1, ("specular", "depth")), (2, ("f0",))):
        rt, _ = make_pair(ren, orc, g, cam, W, H)
        m = rt.cuda_module
        fb = m.get_framebuffer()
        if world > 1:
            m.set_partition(0, world)
        for k in order:
            getattr(fb, "target_" + k).fill_(-7.0)
        m.set_targets_chw(*[None if k in missing else imgs[k] for k in order])
        torch.cuda.synchronize()
        own = torch.ones(H, W, dtype=torch.bool)
        if world > 1:
            owner = importlib.import_module(PKG + ".parallel").tile_owner(W, H, world)  # [mty, mtx]
            own = torch.from_numpy(np.repeat(np.repeat(owner == 0, 16, axis=0), 16, axis=1)[:H, :W].copy())
        for k in order:
            got = getattr(fb, "target_" + k).cpu().reshape(H, W, chans[k])
            want = torch.zeros(H, W, chans[k]) if k in missing else imgs[k].cpu().moveaxis(0, -1)
            assert torch.equal(got[own], want[own]), (size, world, k)
            assert bool((got[~own] == -7.0).all()), (size, world, k)


def test_per_launch_gradient_buffer_never_drops_a_launch(ren, orc, syn):
    """use_grad_delta (egr_set_grad_overwrite): the first grad launch after a fold STORES into the per-launch buffer (stale content is
    overwritten, nobody clears it), a second launch before the fold ADDS (multi-view accumulation, a caller that raised between launch and
    fold), and grad_delta_consumed() makes the next launch store again."""
    W, H = 64, 48
    g = syn.make_scene(2000, "trained", seed=12)
    cam, tg = syn.default_camera(), syn.make_targets(W, H)
    rt, _ = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0))
    m = rt.cuda_module
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    one = m.get_gaussians().grad_flat.clone()
    assert float(one.abs().max()) > 0
    m.use_grad_delta(True)
    gd = m.get_gaussians()
    camera = cam_obj(ren, cam, tg)

    def launch():  # the launch alone (no fold): what a direct cuda_module user does
        m.get_metadata().total_num_calls.zero_()
        m.update_bvh(True)
        m.raytrace()
        torch.cuda.synchronize()

    ren.render(camera, rt)  # sets pose, targets, parameters; its own launch is folded into grad_flat and consumed (all_reduce_grads)
    gd.grad_delta.fill_(123.0)
    scale = float(one.abs().max())
    launch()
    assert float((gd.grad_delta - one).abs().max()) / scale < 1e-5  # stored: the 123s are gone
    launch()
    assert float((gd.grad_delta - 2 * one).abs().max()) / scale < 1e-5  # a second launch before the fold adds
    m.grad_delta_consumed()
    launch()
    assert float((gd.grad_delta - one).abs().max()) / scale < 1e-5  # consumed: stores again


def test_strands_do_not_change_results(ren, orc, syn):
    """The image slices traced on separate HIP streams (egr_set_strands) are independent: the images are bit-identical
    with 1 or 2 strands, the gradients agree to float-atomic reordering."""
    W, H = 96, 64
    g = syn.make_scene(3000, "trained", seed=9)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt, _ = make_pair(ren, orc, g, cam, W, H)
    m = rt.cuda_module
    res = {}
    try:
        m.set_strands(2)
    except RuntimeError:
        pytest.skip("context created with EGR_STRANDS=1")
    for s in (1, 2):
        m.set_strands(s)
        m.get_metadata().total_num_calls.zero_()  # same jitter / GGX random stream for both launches
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        img = hip_outputs(rt)
        m.get_metadata().total_num_calls.zero_()
        run_grad(ren, rt, cam_obj(ren, cam, tg))
        res[s] = (img, hip_grads(rt), m.get_counters()[:9])
    for k in OUT_KEYS:
        assert np.array_equal(res[1][0][k], res[2][0][k]), k
    assert list(res[1][2]) == list(res[2][2])
    for k in GRAD_KEYS:
        assert np.abs(res[1][1][k] - res[2][1][k]).max() / (np.abs(res[1][1][k]).max() + 1e-30) < 1e-4, k
    with pytest.raises(RuntimeError):
        m.set_strands(0)
    with pytest.raises(RuntimeError):
        m.set_strands(99)


def test_team_help_changes_the_list_order_only(ren, orc, syn):
    """egr_set_team_help(1): waves without tiles walk (ray, node) pairs their team mates offer (several waves on one heavy tile). The SET of
    candidates of every ray stays what it was - both per-pixel statistics and every counter equal the run without help - only the order in
    which they enter the ray's list may change - and that order reaches no output: the depth selection orders by (t, list index), so only EXACT
    depth ties could composite in another order, and the total transmittance is an fp64 product rounded once (a small image leaves most waves of every team without a tile, so help is the rule here, not the exception). The switch also
    takes the BACKWARD chain's team build (a partition's under-filled rank gets it without asking: its waves without tiles take batches of
    their team mates' bounce hits): the gradients of the runs with help are compared with the run without."""
    W, H = 96, 64
    g = syn.make_scene(4000, "trained", seed=21)
    cam = syn.default_camera()
    tg = syn.make_targets(W, H)
    rt, _ = make_pair(ren, orc, g, cam, W, H)  # (the run without help is the one the other tests hold against the oracle)
    m = rt.cuda_module
    res = {}
    for help_on in (False, True, True):
        m.set_team_help(help_on)
        m.get_metadata().total_num_calls.zero_()  # same jitter / GGX random stream for all launches
        with torch.no_grad():
            rt(cam_obj(ren, cam))
        img = hip_outputs(rt)
        st = m.get_stats()
        stats = (st.num_traversed_per_pixel.cpu().numpy().copy(), st.num_accumulated_per_pixel.cpu().numpy().copy())
        m.get_metadata().total_num_calls.zero_()
        run_grad(ren, rt, cam_obj(ren, cam, tg))
        res.setdefault(help_on, []).append((img, hip_grads(rt), list(m.get_counters()[:9]), stats, int(m.get_counters()[11])))
    m.set_team_help(False)
    ref = res[False][0]
    worst = 0.0
    for run in res[True]:
        assert run[4] == 0 and run[2] == ref[2], (run[2], ref[2])
        assert np.array_equal(run[3][0], ref[3][0]) and np.array_equal(run[3][1], ref[3][1])  # candidates counted / hits composited per pixel
        for k in OUT_KEYS:
            d = np.abs(run[0][k] - ref[0][k]).max() / (np.abs(ref[0][k]).max() + 1e-30)
            worst = max(worst, d)
            # (round 5: the total transmittance is an fp64 product rounded once, so the list order no longer reaches any output: without exact depth ties -
            # this scene has none - the images of a run with help are the images of the run without, bit for bit)
            assert np.array_equal(run[0][k], ref[0][k]), (k, d)
        for k in GRAD_KEYS:
            assert np.abs(run[1][k] - ref[1][k]).max() / (np.abs(ref[1][k]).max() + 1e-30) < 1e-4, k
    print(f"REPORT team help: worst output difference against the run without help {worst:.2e} of the buffer's maximum")


# ------------------------------------------------------------------------------------------------ full size
def test_full_size_properties_1080p_1M(ren, orc, syn):
    """BASELINE config C (1080p, 1M Gaussians): size-independent properties instead of a full oracle run."""
    W, H, N = 1920, 1080, 1_000_000
    g = syn.make_scene(N, "trained", seed=0)
    cam = syn.default_camera()
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, W, H, ppll_forward_size=400_000_000, ppll_backward_size=300_000_000)
    m = rt.cuda_module
    assert m.check_bvh() == 0, m.last_error()
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    fb = m.get_framebuffer()
    c = m.get_counters()
    assert c[11] == 0 and c[0] == W * H
    st = m.get_stats()
    assert int(st.num_traversed_per_pixel.sum().item()) == c[3] + c[4] + c[5]  # a checksum of checksums
    T, Tt = fb.output_transmittance, fb.output_total_transmittance
    # T (depth order) and T_total (list order, partial products) multiply the same factors in different association orders
    assert bool((Tt <= T + 4e-6).all()) and bool((T <= 1.0).all()) and bool((Tt >= 0).all())
    assert float((fb.output_final[0] - fb.output_rgb.sum(0)).abs().max()) < 1e-5  # shaders.cu:150-152
    dn = torch.linalg.norm(fb.output_ray_direction[0], dim=-1)
    alive = dn > 0
    assert float((dn[alive] - 1).abs().max()) < 1e-3 and float(alive.float().mean()) > 0.5
    # idempotence: same call counter -> bit-identical images (no-grad path has no atomics on pixels)
    a = fb.output_rgb.clone()
    m.get_metadata().total_num_calls.sub_(1)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    assert torch.equal(a, fb.output_rgb)
    # (the gradient check against the oracle AT THIS SIZE is test_hip_configs.py::test_config_c_gradient_check_vs_oracle_at_size; here the
    # same cloud at low resolution, strict parity)
    Ws, Hs = 96, 54
    rt2, o = make_pair(ren, orc, g, cam, Ws, Hs, cfg=dict(jitter_primary_rays=0, num_bounces=0), fwd=50_000_000, bwd=50_000_000)
    with torch.no_grad():
        rt2(cam_obj(ren, cam))
    ref = o.raytrace(False)
    assert psnr(hip_outputs(rt2)["output_rgb"][0], ref["output_rgb"][0]) > 70
    # gradients at full size: linearity in the loss weights (doubling every weight doubles every gradient)
    tg = syn.make_targets(W, H)
    camt = cam_obj(ren, cam, tg)
    m.get_config().jitter_primary_rays.fill_(False)
    run_grad(ren, rt, camt)
    g1 = m.get_gaussians().grad_flat.clone()
    assert m.get_counters()[11] == 0
    cfg = m.get_config()
    for k in ("loss_weight_diffuse", "loss_weight_specular", "loss_weight_depth", "loss_weight_normal", "loss_weight_f0", "loss_weight_roughness"):
        getattr(cfg, k).mul_(2.0)
    m.get_metadata().total_num_calls.sub_(1)  # same RNG stream for the bounce sampling
    run_grad(ren, rt, camt)
    g2 = m.get_gaussians().grad_flat.clone()
    n22 = g1.numel()
    w1, w2 = g1[n22 - N:], g2[n22 - N:]  # total_weight does not depend on the loss weights
    assert float((w1 - w2).abs().max()) <= 1e-3 * float(w1.abs().max())
    d1, d2 = g1[: n22 - N], g2[: n22 - N]
    assert float((d2 - 2 * d1).abs().max()) <= 2e-3 * float(d1.abs().max())


def test_composited_hits_are_capped_at_99_batches_of_16(ren, orc, syn):
    """forward_pass.cu:55 runs at most MAX_ITERATIONS = 99 selection rounds of BUFFER_SIZE = 16 hits (flags.h:15-16): a ray composites at
    most 1584 hits per step, everything behind them only counts in T_total. 2500 nearly transparent gaussians in a row along the
    view axis (distinct depths, threshold 0 so that nothing stops the ray earlier)."""
    W = H = 8
    n = 2500
    g = {"mean": np.stack([2.0 + 0.01 * np.arange(n), np.zeros(n), np.zeros(n)], 1), "scale": np.full((n, 3), np.log(1.5)),
         "rotation": np.tile(np.array([1.0, 0, 0, 0]), (n, 1)), "opacity": np.full((n, 1), np.log(0.01 / 0.99)),
         "rgb": np.random.default_rng(0).uniform(0.1, 0.9, (n, 3)), "normal": np.tile(np.array([-1.0, 0, 0]), (n, 1)),
         "f0": np.full((n, 3), 0.04), "roughness": np.full((n, 1), 0.3)}
    g = {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in g.items()}
    cam = syn.plus_x_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0, transmittance_threshold=0.0), fwd=20_000_000, bwd=20_000_000)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    ha = rt.cuda_module.get_stats().num_accumulated_per_pixel.cpu().numpy()
    assert int(ref["num_accumulated"].max()) == 1584 and int(ha.max()) == 1584
    bad, nbad = mismatch_list(ha, ref["num_accumulated"])
    report("hit_cap", mismatching_pixels=nbad, first=bad, capped_pixels=int((ha == 1584).sum()))
    assert nbad == 0, bad
    capped = (ha == 1584)
    assert bool(np.all(out["output_total_transmittance"][0][capped] < out["output_transmittance"][0][capped]))  # the rest is in T_total only
    for k in ("output_rgb", "output_depth", "output_transmittance", "output_total_transmittance"):
        assert psnr(out[k], ref[k]) > 100, k
    # gradients of the capped rays: the arena holds 198 blocks of 8 hits per tile
    tg = syn.make_targets(W, H)
    run_grad(ren, rt, cam_obj(ren, cam, tg))
    refg = o.raytrace(True, targets=tg)
    gr = hip_grads(rt)
    # (the stacked gaussians are axis-aligned copies: their rotation gradient cancels to exactly 0 in the oracle's summation order; the
    # kernel's order - neighbour pre-sums, LDS table, atomics - leaves rounding residue, so a tensor that is zero in the reference is
    # held to 1e-7 of the largest gradient of the launch instead of to itself)
    floor = 1e-4 * max(float(np.abs(refg[k]).max()) for k in GRAD_KEYS)  # (x the 1e-3 below = 1e-7 of the largest gradient: fp32 rounding of the cancelling terms)
    for k in GRAD_KEYS:
        assert np.abs(gr[k] - refg[k]).max() / max(float(np.abs(refg[k]).max()), floor, 1e-30) < 1e-3, k
    assert rt.cuda_module.get_counters()[11] == 0


def test_axis_parallel_rays_are_pruned_like_their_neighbours(ren, orc, syn):
    """A direction component of EXACTLY zero makes both slab distances of that axis inf - inf: the quantised box test then drops the
    axis (conservative, so the images stay right) and the ray overlaps every box along its line - a packet walks what any of its rays
    overlaps, and one such primary ray per twenty launches of the bench made its tile take 12 ms (found through the rocprofv3 average
    of round 3). With an odd image size and an axis-aligned camera the centre column / row have such components by construction: they
    must render like the oracle AND evaluate about as many records as the columns next to them."""
    W, H = 65, 33
    g = syn.random_blob_scene(6000, seed=4, extent=1.5, depth_range=(1.0, 6.0), scale_range=(0.02, 0.2))
    cam = syn.plus_x_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H, cfg=dict(jitter_primary_rays=0, num_bounces=0), fwd=50_000_000, bwd=10_000_000)
    with torch.no_grad():
        rt(cam_obj(ren, cam))
    d = o.primary_rays(jitter=False)
    assert d[H // 2, W // 2, 1] == 0.0 and d[H // 2, W // 2, 2] == 0.0 and np.all(d[:, W // 2, 1] == 0.0)  # the construction holds
    ref = o.raytrace(False)
    out = hip_outputs(rt)
    assert psnr(out["output_rgb"], ref["output_rgb"]) > 90 and psnr(out["output_total_transmittance"], ref["output_total_transmittance"]) > 90
    tr = rt.cuda_module.get_stats().num_traversed_per_pixel.cpu().numpy().astype(np.float64)
    col, nb = tr[:, W // 2].mean(), 0.5 * (tr[:, W // 2 - 1].mean() + tr[:, W // 2 + 1].mean())
    row, nbr = tr[H // 2, :].mean(), 0.5 * (tr[H // 2 - 1, :].mean() + tr[H // 2 + 1, :].mean())
    report("axis_parallel_rays", centre_column_over_neighbours=round(col / nb, 2), centre_row_over_neighbours=round(row / nbr, 2))
    assert col < 1.5 * nb and row < 1.5 * nbr, (col, nb, row, nbr)


def test_lean_division_and_square_root_are_ieee_on_their_domain(ren):
    """egr_div_rn / egr_sqrt_rn (csrc/egr_device.hpp) replace every `/` and sqrtf of the candidate test, the step epilogue and the backward with the
    compiler's own correction steps WITHOUT its range scaling (v_div_scale / v_div_fixup, the 2^32 pre-scaling of the radicand). On the accepted domain
    (include/egr_raytracer.h: egr_debug_lean_arith) the results are IEEE's, bit for bit - held against numpy's float32 division / square root over 4M random
    operand pairs spanning 2^-60 ... 2^60 and a list of hand-picked ones; outside the domain the deviation is documented, and shown here."""
    importlib.import_module(PKG).load_library()
    rng = np.random.default_rng(5)
    n = 1 << 22
    mant = lambda: rng.uniform(1.0, 2.0, n).astype(np.float32)
    a = (mant() * np.exp2(rng.integers(-60, 60, n)).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)).astype(np.float32)
    b = (mant() * np.exp2(rng.integers(-60, 60, n)).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)).astype(np.float32)
    hand_a = np.float32([1, 1, 2, 3, 0, -0.0, 1e-30, 1e30, 0.1, 7, 16777216, 16777217, 1.0000001, 0.99999994, 5e-20, 3.4e20, 1, 2, 4, 1e10])
    hand_b = np.float32([3, 7, 3, 2, 5, 5, 1e-10, 1e10, 0.3, 0.7, 3, 3, 0.99999994, 1.0000001, 7e19, 1.1e-19, 1e-30, 1e30, 2 ** -100, 2 ** 100])
    a = np.concatenate([hand_a, a]).astype(np.float32)
    b = np.concatenate([hand_b, b]).astype(np.float32)
    q2, _ = torch.ops.egr.debug_lean_arith(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    ok = np.isfinite(a / b) & ((np.abs(a / b) >= np.float32(2.0 ** -126)) | (a == 0))  # quotient normal (or exactly 0): the accepted domain
    got = q2.cpu().numpy()
    bad = np.flatnonzero(ok & (got.view(np.uint32) != (a / b).view(np.uint32)) & ~((got == 0) & (a / b == 0)))  # (the sign of a zero quotient is not kept: -0 / 5 gives +0)
    assert bad.size == 0, [(float(a[i]), float(b[i]), float(got[i]), float((a / b)[i])) for i in bad[:8]]
    # square roots of the positive operands
    _, r = torch.ops.egr.debug_lean_arith(torch.from_numpy(np.abs(a)).cuda(), torch.from_numpy(np.abs(b)).cuda())
    rr, want = r.cpu().numpy(), np.sqrt(np.abs(a))
    okr = (np.abs(a) >= np.float32(2.0 ** -100)) | (a == 0)
    badr = np.flatnonzero(okr & (rr.view(np.uint32) != want.view(np.uint32)))
    assert badr.size == 0, [(float(a[i]), float(rr[i]), float(want[i])) for i in badr[:8]]
    # outside the domain (documented): b = 0 and b = inf give NaN where IEEE gives inf / 0 - never a finite wrong value
    ea, eb = np.float32([1, 1, 0, np.inf]), np.float32([0, np.inf, 0, 2])
    eq, _ = torch.ops.egr.debug_lean_arith(torch.from_numpy(ea).cuda(), torch.from_numpy(eb).cuda())
    eq = eq.cpu().numpy()
    report("lean_arith_outside_the_domain", one_over_zero=float(eq[0]), one_over_inf=float(eq[1]), zero_over_zero=float(eq[2]), inf_over_two=float(eq[3]))
    assert all((not np.isfinite(x)) or x == 0.0 for x in eq), eq
