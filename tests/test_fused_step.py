"""the fused host step (csrc/step.hip, trainer.py) against the reference's own sequence of torch calls."""
import ast
import importlib
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch = pytest.importorskip("torch")
PKG = "editable-gaussian-reflections_amd"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _expon_lr():
    # trainer.py loads the native library on import (GPU product); the schedule itself is pure numpy: load it without that
    src = open(os.path.join(ROOT, PKG, "trainer.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "expon_lr")
    ns = {"np": np, "math": math}
    exec(compile(ast.Module([fn], []), "trainer.expon_lr", "exec"), ns)
    return ns["expon_lr"]


def test_lr_schedule_matches_reference_vectors():
    """tests/golden/expon_lr.npz was produced by the reference's get_expon_lr_func (make_expon_lr_vectors.py)."""
    f = _expon_lr()
    z = np.load(os.path.join(GOLD, "expon_lr.npz"))
    for k, c in enumerate(z["cases"]):
        kw = eval(str(c))
        got = np.array([f(int(s), **kw) for s in z["steps"]])
        np.testing.assert_allclose(got, z[f"lr{k}"], rtol=1e-12, atol=0)


LRS = dict(xyz=1.6e-4, normal=1e-3, roughness=2e-3, f0=2e-3, f_dc=2.5e-3, opacity=2.5e-2, scaling=5e-3, rotation=1e-3)


def reference_sequence(pc, rtg, opt, scale_decay):
    """train.py:224-254 + gaussian_raytracer.py:41-62 with stock torch calls."""
    with torch.no_grad():
        pc._xyz.grad.add_(rtg.mean.grad), pc._opacity.grad.add_(rtg.opacity.grad), pc._scaling.grad.add_(rtg.scale.grad)
        pc._rotation.grad.add_(rtg.rotation.grad), pc._diffuse.grad.add_(rtg.rgb.grad), pc._normal.grad.add_(rtg.normal.grad)
        pc._roughness.grad.add_(rtg.roughness.grad), pc._f0.grad.add_(rtg.f0.grad)
        if scale_decay < 1.0:
            pc._scaling.copy_(torch.log(torch.exp(pc._scaling) * scale_decay))
        opt.step()
        opt.zero_grad(set_to_none=False)
        for t in (rtg.rgb, rtg.opacity, rtg.scale, rtg.rotation, rtg.mean, rtg.normal, rtg.roughness, rtg.f0):
            t.grad.zero_()
        pc._diffuse.data.clamp_(min=0.0), pc._roughness.data.clamp_(min=0.0, max=1.0), pc._f0.data.clamp_(min=0.0, max=1.0)
        rtg.scale.copy_(pc._scaling), rtg.rotation.copy_(pc._rotation), rtg.mean.copy_(pc._xyz), rtg.opacity.copy_(pc._opacity)
        rtg.rgb.copy_(pc._diffuse), rtg.normal.copy_(pc._normal), rtg.roughness.copy_(pc._roughness), rtg.f0.copy_(pc._f0)


@pytest.mark.gpu
@pytest.mark.parametrize("scale_decay", [1.0, 0.999])
def test_fused_step_matches_torch_adam(scale_decay):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    ren = importlib.import_module(PKG + ".renderer")
    syn = importlib.import_module(PKG + ".synthetic")
    tr = importlib.import_module(PKG + ".trainer")
    N = 20000
    g = syn.make_scene(N, "trained", seed=4)
    pcs, rts = [], []
    for _ in range(2):
        pc = ren.GaussianParams(g)
        pcs.append(pc), rts.append(ren.GaussianRaytracer(pc, 32, 32))
    (pa, pb), (ra, rb) = pcs, rts
    groups = [{"params": [getattr(pb, attr)], "lr": LRS[name], "name": name} for name, attr, _ in tr.GROUPS]
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15, betas=(0.9, 0.999))
    fused = tr.FusedTrainStep(pa, ra, LRS, scale_decay=scale_decay)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for it in range(6):
        for _, attr, rtname in tr.GROUPS:  # same synthetic gradients on both sides: raytracer gradients + a model-side part
            shape = getattr(pa, attr).shape
            d_rt = torch.randn(shape, device="cuda", generator=gen) * 10.0 ** float(torch.randint(-6, 1, (1,)).item())
            d_model = torch.randn(shape, device="cuda", generator=gen) * 1e-3 * (it % 2)
            for pc, rt in ((pa, ra), (pb, rb)):
                getattr(rt.cuda_module.get_gaussians(), rtname).grad.copy_(d_rt)
                getattr(pc, attr).grad.copy_(d_model)
        fused.step()
        reference_sequence(pb, rb.cuda_module.get_gaussians(), opt, scale_decay)
        ga, gb = ra.cuda_module.get_gaussians(), rb.cuda_module.get_gaussians()
        for name, attr, rtname in tr.GROUPS:
            a, b = getattr(pa, attr), getattr(pb, attr)
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, name, float((a - b).abs().max()))
            assert torch.equal(getattr(ga, rtname), a), (it, name)  # export
            assert float(getattr(ga, rtname).grad.abs().max()) == 0.0 and float(a.grad.abs().max()) == 0.0  # both zero_grads
            st = opt.state[b]
            for mine, ref in ((fused.exp_avg[name], st["exp_avg"]), (fused.exp_avg_sq[name], st["exp_avg_sq"])):
                # fp32 round-off only (the lerp cancels: compare against the tensor's scale, not element by element)
                assert float((mine - ref).abs().max()) <= 1e-6 * float(ref.abs().max()) + 1e-30, (it, name)
    assert float(pa._diffuse.min()) >= 0.0 and float(pa._roughness.max()) <= 1.0 and float(pa._f0.min()) >= 0.0


@pytest.mark.gpu
def test_reset_state_keeps_the_step_count_like_replace_tensor_to_optimizer():
    """scene/gaussian_model.py:464-476 zeroes exp_avg / exp_avg_sq of one group and re-attaches the SAME state dict, so torch's
    per-tensor `step` - the bias correction - carries on. The fused step must do the same: after 5 steps and a reset of "opacity"
    the next update equals torch.optim.Adam's on a state whose moments were zeroed in place."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU fallback")
    ren = importlib.import_module(PKG + ".renderer")
    syn = importlib.import_module(PKG + ".synthetic")
    tr = importlib.import_module(PKG + ".trainer")
    N = 4000
    g = syn.make_scene(N, "trained", seed=6)
    pa, pb = ren.GaussianParams(g), ren.GaussianParams(g)
    ra, rb = ren.GaussianRaytracer(pa, 32, 32), ren.GaussianRaytracer(pb, 32, 32)
    opt = torch.optim.Adam([{"params": [getattr(pb, attr)], "lr": LRS[name], "name": name} for name, attr, _ in tr.GROUPS], lr=0.0, eps=1e-15)
    fused = tr.FusedTrainStep(pa, ra, LRS)
    gen = torch.Generator(device="cuda").manual_seed(7)
    for it in range(8):
        if it == 5:  # opacity reset (train.py resets opacities every opacity_reset_interval): new values, zero moments, same step count
            new_opa = torch.full_like(pa._opacity, -2.0)
            pa._opacity.copy_(new_opa), pb._opacity.data.copy_(new_opa)
            fused.reset_state("opacity")
            st = opt.state[pb._opacity]
            st["exp_avg"].zero_(), st["exp_avg_sq"].zero_()  # what replace_tensor_to_optimizer leaves behind (`step` untouched)
            assert int(st["step"]) == 5
        for _, attr, rtname in tr.GROUPS:
            d = torch.randn(getattr(pa, attr).shape, device="cuda", generator=gen) * 1e-2
            for pc, rt in ((pa, ra), (pb, rb)):
                getattr(rt.cuda_module.get_gaussians(), rtname).grad.copy_(d)
        before = pa._opacity.clone()
        fused.step()
        reference_sequence(pb, rb.cuda_module.get_gaussians(), opt, 1.0)
        for name, attr, _ in tr.GROUPS:
            a, b = getattr(pa, attr), getattr(pb, attr)
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, name, float((a - b).abs().max()))
        if it == 5:  # the first update after the reset is (1-b1)/sqrt(1-b2) * sqrt(1-b2^6)/(1-b1^6) ~ 0.52 lr, not lr (a restarted count)
            ratio = float(((pa._opacity - before).abs() / LRS["opacity"]).median())
            want = (1 - 0.9) / (1 - 0.9 ** 6) / math.sqrt((1 - 0.999) / (1 - 0.999 ** 6))
            assert abs(ratio - want) < 1e-3 * want, (ratio, want)


@pytest.mark.gpu
def test_fused_step_full_size_timing():
    ren = importlib.import_module(PKG + ".renderer")
    syn = importlib.import_module(PKG + ".synthetic")
    tr = importlib.import_module(PKG + ".trainer")
    import time
    N = 1_000_000
    g = syn.make_scene(N, "trained", seed=0)
    pc = ren.GaussianParams(g)
    rt = ren.GaussianRaytracer(pc, 64, 64)
    fused = tr.FusedTrainStep(pc, rt, LRS, scale_decay=0.9999)
    opt = torch.optim.Adam([{"params": [getattr(pc, attr)], "lr": LRS[name]} for name, attr, _ in tr.GROUPS], lr=0.0, eps=1e-15)
    rtg = rt.cuda_module.get_gaussians()

    def timed(f, reps=20):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    t_fused = timed(fused.step)
    t_torch = timed(lambda: reference_sequence(pc, rtg, opt, 0.9999))
    print(f"host step at N=1M: fused {t_fused:.3f} ms vs stock torch sequence {t_torch:.3f} ms")
    assert t_fused < t_torch
    with pytest.raises(RuntimeError):
        torch.ops.egr.fused_adam_step([pc._xyz], [pc._xyz.grad], [], [], [], [], [1.0], [-math.inf], [math.inf], [1.0], 1, 0.9, 0.999, 1e-15)
    with pytest.raises(RuntimeError):  # a per-group step list of the wrong length is rejected, not silently ignored
        torch.ops.egr.fused_adam_step([pc._xyz], [pc._xyz.grad], [pc._xyz], [pc._xyz.grad], [pc._xyz], [pc._xyz], [1.0], [-math.inf], [math.inf], [1.0], 1, 0.9,
                                      0.999, 1e-15, [1, 2])
