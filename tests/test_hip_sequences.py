"""GPU tests of the CALL SEQUENCES behind BASELINE configs 4 and 5 (the datasets - shiny_kitchen, neural_catacaustics - are not
in this image, so the synthetic room stands in; the tracer-visible call patterns are the reference's):

  config 4  train.py:211-263 at 1080p with the tile split + gradient exchange, prune / rebuild at the interval, far-field growth
            mid-interval - two ranks sharing the box's one GPU, compared with an unpartitioned run (tests/config4_worker.py)
  config 5  the editing / viewer / multi-sample render patterns against the CPU oracle (below)
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from hip_common import GRAD_KEYS, PKG, cam_obj, generic_targets, hip_grads, hip_outputs, make_pair, psnr, ren, report, run_grad  # noqa: F401,E402


@pytest.mark.timeout(1500)
def test_config4_substitute_two_ranks_at_size():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "tests", "config4_worker.py")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1400, cwd=ROOT)
    for line in r.stdout.splitlines():
        if line.startswith("CONFIG4"):
            print("REPORT " + line, flush=True)
    assert r.returncode == 0 and "CONFIG4_OK" in r.stdout, r.stdout[-6000:]


def _current(pc):
    """the model's raw tensors as the oracle's input dict"""
    f = lambda t: t.detach().cpu().numpy()
    return dict(mean=f(pc._xyz), opacity=f(pc._opacity), scale=f(pc._scaling), rotation=f(pc._rotation), rgb=f(pc._diffuse), normal=f(pc._normal),
                roughness=f(pc._roughness), f0=f(pc._f0))


def test_config5_substitute_editing_and_viewer_sequences_vs_oracle(ren, orc, syn):
    """BASELINE config 5 (real multi-bounce scene + editing) as its tracer-visible call patterns on the synthetic room, every frame
    against the CPU oracle driven through the same sequence (reference defaults: jitter on, two bounces):
      1. object REMOVAL = opacity_raw * 0 - 1e8 (scene/editable_gaussian_model.py:325-328), first seen through the LIVE opacity only
         (no_grad frame, no refit: alpha = 0 candidates are still composited), then with force_update_bvh = is_dirty
         (gaussian_viewer.py:341), where sigma = 0 masks the instances out;
      2. DUPLICATION = concatenated tensors + rebuild_bvh() (:284-322, gaussian_viewer.py:327-330);
      3. live config edits between frames with backup / restore (gaussian_viewer.py:335-344): num_bounces acts at once,
         global_scale_factor only with the next refit;
      4. multi-sample render: accumulate_samples, reset_accumulators(), spp launches (render.py:195-209), reset, again; denoise()."""
    W, H, N = 96, 64, 6000
    g = syn.make_scene(N, "trained", seed=23)
    cam = syn.default_camera()
    rt, o = make_pair(ren, orc, g, cam, W, H)
    pc, m = rt.pc, rt.cuda_module
    camera = cam_obj(ren, cam)
    levels = {}

    def frame(tag, force=False, bar=50.0):
        o.set_gaussians(_current(pc))  # __call__ exports the live parameters on every frame
        if force:
            o.update_bvh()
        with torch.no_grad():
            pkg = ren.render(camera, rt, targets_available=False, force_update_bvh=force)
        ref = o.raytrace(False)
        out = hip_outputs(rt)
        lv = {k: psnr(out[k], ref[k]) for k in ("output_rgb", "output_final", "output_normal", "output_total_transmittance")}
        lv["output_depth"] = psnr(out["output_depth"] / 4.0, ref["output_depth"] / 4.0)  # (the room is 4 units across: depth on the [0, 1] scale PSNR assumes)
        lv["rgb_step1"] = psnr(out["output_rgb"][1], ref["output_rgb"][1])
        worst = min(lv, key=lv.get)
        levels[tag] = f"{lv[worst]:.1f}({worst.replace('output_', '')})"
        assert min(lv.values()) > bar, (tag, lv)
        assert m.get_counters()[11] == 0
        assert pkg.final.shape == (1, 3, H, W)
        return out

    base = frame("baseline")
    mean = pc._xyz
    sphere = lambda c: ((mean - torch.tensor(c, device="cuda")).norm(dim=1) < 0.56)
    # 1. removal
    sel = sphere((1.0, 0.8, -1.0))
    assert 50 < int(sel.sum()) < N // 4
    pc._opacity[sel] *= 0.0
    pc._opacity[sel] -= 100000000.0
    live = frame("removed_live_opacity")
    assert np.abs(live["output_rgb"][0] - base["output_rgb"][0]).max() > 0.05  # the sphere is gone from the primary image
    refit = frame("removed_refit", force=True)
    assert psnr(refit["output_rgb"][0], live["output_rgb"][0]) > 20  # the same picture either way (alpha = 0 hits carry no weight) up to another frame's jitter
    # 2. duplication
    sel2 = sphere((1.0, -0.8, -1.0))
    k = int(sel2.sum())
    off = torch.tensor([0.0, 0.0, 0.9], device="cuda")
    for name in pc._NAMES:
        t = getattr(pc, name)
        add = t[sel2].clone() + (off if name == "_xyz" else 0.0)
        setattr(pc, name, torch.cat((t, add), 0).contiguous())
    rt.rebuild_bvh()
    assert m.get_gaussians().mean.shape[0] == N + k and m.check_bvh() == 0
    o.set_gaussians(_current(pc))
    o.update_bvh()
    dup = frame("duplicated")
    assert np.abs(dup["output_rgb"][0] - refit["output_rgb"][0]).max() > 0.05
    # 3. live config edits with backup / restore
    cfg = m.get_config()
    bkp = (cfg.num_bounces.clone(), cfg.global_scale_factor.clone())
    cfg.num_bounces.copy_(torch.tensor([1], dtype=torch.int32))
    cfg.global_scale_factor.copy_(torch.tensor([1.3]))
    o.set_config(num_bounces=1, global_scale_factor=1.3)
    one = frame("one_bounce_scale_pending")
    assert float(np.abs(one["output_rgb"][2]).max()) == 0.0 and psnr(one["output_rgb"][0], dup["output_rgb"][0]) > 20  # transforms not refitted yet (other jitter)
    scaled = frame("one_bounce_scaled", force=True)
    assert np.abs(scaled["output_rgb"][0] - one["output_rgb"][0]).max() > 0.01
    cfg.num_bounces.copy_(bkp[0]), cfg.global_scale_factor.copy_(bkp[1])
    o.set_config(num_bounces=2, global_scale_factor=1.0)
    frame("restored", force=True)
    # 4. multi-sample render
    cfg.accumulate_samples.copy_(torch.tensor([True]))
    o.set_config(accumulate_samples=1)
    fb = m.get_framebuffer()
    for rounds, spp in enumerate((8, 4)):
        m.reset_accumulators()
        o.reset_accumulators()
        assert int(fb.accumulated_sample_count) == 0 and float(fb.accumulated_rgb.abs().max()) == 0.0
        for s in range(spp):
            out = frame(f"spp_round{rounds}_sample{s}")
            assert int(fb.accumulated_sample_count) == s + 1
    one_sample = out
    m.denoise()
    torch.cuda.synchronize()
    den = fb.output_denoised.cpu().numpy()
    assert den.shape == (1, H, W, 3) and np.isfinite(den).all() and psnr(den, one_sample["output_final"]) > 20  # (stand-in filter: parity unpinned)
    cfg.accumulate_samples.copy_(torch.tensor([False]))
    report("config5_sequences", **levels)


@pytest.mark.parametrize("scene", ["trained_1M", "init_1M", "blobs_200k"])
def test_reference_default_capacities_at_size(ren, syn, scene):
    """A drop-in caller never passes capacities: `GaussianRaytracer(pc, W, H)` -> make_raytracer's 180M / 120M list entries
    (/root/reference/editable_gauss_refl/__init__.py:19-20 = 6.5 + 4.3 GB upstream). One training iteration at 1080p on both 1M
    clouds of the bench must end with status 0 inside those budgets. The 200k-blob stress scene (thousands of candidates per ray) is
    beyond the REFERENCE's own budget at this resolution - its forward list would take `accepted` entries (egr_counters.accepted:
    one per accepted candidate, shaders.cu:74) where 180M exist, and per_pixel_linked_list.h:30-42 does not check - so there the
    contract is: flagged, never silent, everything finite. The context's own device memory is reported (egr_counters.device_bytes)."""
    W, H = 1920, 1080
    if scene == "blobs_200k":
        g, cam = syn.random_blob_scene(200_000, seed=1, extent=2.0, depth_range=(1.0, 8.0), scale_range=(0.002, 0.3)), syn.plus_x_camera()
    else:
        g, cam = syn.make_scene(1_000_000, scene.split("_")[0], seed=0), syn.default_camera()
    tg = syn.make_targets(W, H)
    rt = ren.GaussianRaytracer(ren.GaussianParams(g), W, H)  # NO capacity arguments
    m = rt.cuda_module
    camt = cam_obj(ren, cam, tg)
    for _ in range(2):
        run_grad(ren, rt, camt)
    c = m.get_counters()
    report("default_capacities_" + scene, status=c[11], device_GB=round(c[14] / 1e9, 2), arena_blocks=f"{c[15]}/{c[16]}", ext_blocks=f"{c[17]}/{c[18]}",
           rays=list(c[0:3]), composited_per_ray=[round(c[6 + i] / max(c[i], 1), 1) for i in range(3)])
    accepted = sum(c[19:22])
    report("default_capacities_" + scene + "_reference_list_entries", accepted=accepted, of=180_000_000)
    if accepted <= 180_000_000:  # fits the reference's forward list: must fit here
        assert c[11] == 0, ("capacity overflow with the reference's default sizes", c[11], c[15], c[16], c[17], c[18])
    else:
        assert scene == "blobs_200k" and (c[11] & 1) == 1, (accepted, c[11])  # upstream: out-of-bounds writes; here: dropped candidates, flagged
    assert bool(torch.isfinite(m.get_gaussians().grad_flat).all()) and c[0] == W * H
    assert c[14] < 20e9  # the reference's own lists take 10.8 GB at these sizes; ours must stay in that class
