"""Host step of one training iteration as ONE HIP launch.

Mirrors what the reference does around `render()` in train.py:217-254 with ~60 small torch kernels:
gradient import (renderer/gaussian_raytracer.py:53-62), scale decay (train.py:224-226), `gaussians.optimizer.step()` =
torch.optim.Adam(eps=1e-15, betas=(beta_1, beta_2)) over the eight parameter groups of scene/gaussian_model.py:296-338,
`zero_grad` of the model and of the raytracer (train.py:248-249), the three clamps (train.py:251-254) and the parameter
export the next `GaussianRaytracer.__call__` would do (gaussian_raytracer.py:41-50). `expon_lr` restates
utils/general_utils.py:31-60 (pinned by tests/golden/expon_lr.npz, generated from the reference's own function).
"""
import importlib
import math

import torch

importlib.import_module(__package__).load_library()

# (optimizer group name, model attribute, raytracer tensor) in the order of gaussian_model.py:296-326
GROUPS = (("xyz", "_xyz", "mean"), ("normal", "_normal", "normal"), ("roughness", "_roughness", "roughness"), ("f0", "_f0", "f0"),
          ("f_dc", "_diffuse", "rgb"), ("opacity", "_opacity", "opacity"), ("scaling", "_scaling", "scale"), ("rotation", "_rotation", "rotation"))
CLAMPS = {"f_dc": (0.0, math.inf), "roughness": (0.0, 1.0), "f0": (0.0, 1.0)}  # train.py:251-254


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """The xyz learning-rate schedule (behaviour of utils/general_utils.py:31-60, pinned by tests/golden/expon_lr.npz):
    geometric interpolation lr_init -> lr_final over max_steps, times a warm-up factor that eases from lr_delay_mult to 1
    along a quarter sine over the first lr_delay_steps. A negative step or two zero rates switch the group off."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    frac = min(max(step / max_steps, 0.0), 1.0)
    rate = lr_init * (lr_final / lr_init) ** frac  # = exp((1 - frac) ln lr_init + frac ln lr_final)
    if lr_delay_steps > 0:
        ease = math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        rate *= lr_delay_mult + (1.0 - lr_delay_mult) * ease
    return float(rate)


class FusedTrainStep:
    """`step()` = import + scale decay + Adam + clamps + both zero_grads + export, one kernel over all eight groups.

    pc: model with the reference's raw parameter attributes (each with a `.grad`); raytracer: renderer.GaussianRaytracer;
    lrs: {group name: learning rate}; xyz_schedule: kwargs of `expon_lr` (the reference only schedules xyz,
    gaussian_model.py:349-355)."""

    def __init__(self, pc, raytracer, lrs, beta1=0.9, beta2=0.999, eps=1e-15, scale_decay=1.0, xyz_schedule=None):
        self.pc, self.rt = pc, raytracer
        self.lrs = {name: float(lrs.get(name, 0.0)) for name, _, _ in GROUPS}
        self.beta1, self.beta2, self.eps, self.scale_decay = beta1, beta2, eps, scale_decay
        self.xyz_schedule = xyz_schedule
        self.steps = 0
        self.exp_avg = {name: torch.zeros_like(getattr(pc, attr)) for name, attr, _ in GROUPS}
        self.exp_avg_sq = {name: torch.zeros_like(getattr(pc, attr)) for name, attr, _ in GROUPS}
        # the fused kernel adds the raytracer-side gradients itself: GaussianRaytracer.__call__ must not import them as well
        # (it would count them twice: g = model_grad + 2 * rt_grad)
        raytracer.import_grads = False

    # ---- optimizer surgery, mirroring scene/gaussian_model.py (the moments follow the parameters through topology changes)
    @torch.no_grad()
    def prune(self, keep_mask):
        """gaussian_model.py `_prune_optimizer`: keep the rows of both moments where `keep_mask` is True. The caller prunes the
        parameters themselves (`pc.prune_points(~keep_mask)`) and calls raytracer.rebuild_bvh().
        train.py:238-249 prunes BETWEEN render() and optimizer.step(): upstream the raytracer gradients of that iteration were
        already added to the old parameters' `.grad`, which prune_points replaces by zeros, so the step that follows sees a zero
        gradient. Here the import happens inside the fused step, so the iteration's raytracer gradients are dropped now
        (resize() would otherwise keep their first rows, misaligned with the pruned model)."""
        for name in self.exp_avg:
            self.exp_avg[name] = self.exp_avg[name][keep_mask].contiguous()
            self.exp_avg_sq[name] = self.exp_avg_sq[name][keep_mask].contiguous()
        self.rt.zero_grad()

    @torch.no_grad()
    def extend(self, n_new):
        """gaussian_model.py `cat_tensors_to_optimizer` (add_farfield_points): new rows start with zero moments."""
        for name in self.exp_avg:
            z = self.exp_avg[name].new_zeros((n_new,) + tuple(self.exp_avg[name].shape[1:]))
            self.exp_avg[name] = torch.cat([self.exp_avg[name], z], 0)
            self.exp_avg_sq[name] = torch.cat([self.exp_avg_sq[name], z.clone()], 0)

    @torch.no_grad()
    def reset_state(self, name):
        """gaussian_model.py:464-476 `replace_tensor_to_optimizer`: the group's parameter was replaced, both moments restart at zero.
        The stored state object - and with it torch's `step` count, hence the bias correction - is re-attached unchanged, so the
        first update after a reset is lr * m_hat / sqrt(v_hat) with the OLD step count (about 3.16 lr sign(g) late in training,
        not lr sign(g))."""
        attr = {n: a for n, a, _ in GROUPS}[name]
        self.exp_avg[name] = torch.zeros_like(getattr(self.pc, attr))
        self.exp_avg_sq[name] = torch.zeros_like(getattr(self.pc, attr))

    def update_learning_rate(self, iteration):  # gaussian_model.py:349-355
        if self.xyz_schedule is not None:
            self.lrs["xyz"] = expon_lr(iteration, **self.xyz_schedule)
        return self.lrs["xyz"]

    @torch.no_grad()
    def step(self):
        g = self.rt.cuda_module.get_gaussians()
        self.steps += 1  # one count for all groups: none of the reference's optimizer surgery touches torch's per-tensor `step`
        params = [getattr(self.pc, attr) for _, attr, _ in GROUPS]
        grads = [getattr(self.pc, attr).grad for _, attr, _ in GROUPS]
        rt_params = [getattr(g, rt) for _, _, rt in GROUPS]
        rt_grads = [getattr(g, rt).grad for _, _, rt in GROUPS]
        torch.ops.egr.fused_adam_step(
            params, grads, rt_params, rt_grads, [self.exp_avg[n] for n, _, _ in GROUPS], [self.exp_avg_sq[n] for n, _, _ in GROUPS],
            [self.lrs[n] for n, _, _ in GROUPS], [CLAMPS.get(n, (-math.inf, math.inf))[0] for n, _, _ in GROUPS],
            [CLAMPS.get(n, (-math.inf, math.inf))[1] for n, _, _ in GROUPS], [self.scale_decay if n == "scaling" else 1.0 for n, _, _ in GROUPS],
            self.steps, self.beta1, self.beta2, self.eps)
