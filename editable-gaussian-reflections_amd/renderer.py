"""Host-side mirror of the reference's L3 caller:
editable_gauss_refl/renderer/gaussian_raytracer.py:10-151 (`GaussianRaytracer`) and
editable_gauss_refl/renderer/gaussian_renderer.py:21-92 (`render`).

Same call sequence, attribute names and tensor conventions, so a reference `GaussianModel` / `Camera` can be
passed unchanged: pose conversion R_blender = -R with column 0 re-negated, set_pose(camera_center, R_blender),
fov = FoVy, ZNEAR/ZFAR env overrides, export order, targets CHW -> HWC, zero targets when absent, update_bvh iff
grad mode or forced, raytrace, optional denoise, gradient import by add_.

Additions for the MI355X build: `rank` / `world_size` (image-tile partition), `all_reduce_grads()` which sums
THIS launch's [22N] gradient contribution over ranks with ONE RCCL all-reduce (`parallel.all_reduce_launch_delta`) and folds it
into the persistent gradient buffer, and `gather_outputs()` which completes the images of a partitioned no-grad render on every
rank with one all-gather (`parallel.ImageGather`).
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import make_raytracer
from .parallel import ImageGather, all_reduce_launch_delta


class GaussianParams:
    """Minimal stand-in for the parts of scene/gaussian_model.py:31 the tracer reads (raw, pre-activation tensors
    with the reference's attribute names). `cfg` carries the nine values pushed into the native config."""

    def __init__(self, g, device="cuda", cfg=None):
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device).contiguous()
        self._xyz, self._opacity, self._scaling, self._rotation = t(g["mean"]), t(g["opacity"]), t(g["scale"]), t(g["rotation"])
        self._diffuse, self._normal, self._roughness, self._f0 = t(g["rgb"]), t(g["normal"]), t(g["roughness"]), t(g["f0"])
        for p in self.parameters():
            p.grad = torch.zeros_like(p)
        self.cfg = cfg or SimpleNamespace(loss_weight_diffuse=5.0, loss_weight_specular=3.0, loss_weight_normal=2.5, loss_weight_depth=2.5,
                                          loss_weight_f0=1.0, loss_weight_roughness=1.0, transmittance_threshold=0.01, alpha_threshold=0.005,
                                          exp_power=3)

    def parameters(self):
        return [self._xyz, self._opacity, self._scaling, self._rotation, self._diffuse, self._normal, self._roughness, self._f0]

    _NAMES = ("_xyz", "_opacity", "_scaling", "_rotation", "_diffuse", "_normal", "_roughness", "_f0")

    @torch.no_grad()
    def prune_points(self, mask):
        """scene/gaussian_model.py:493-531 `prune_points(mask)`: rows with mask == True are REMOVED; every parameter is re-created
        from the kept rows with a fresh zero `.grad` (the gradient of the iteration in flight is dropped, like upstream)."""
        keep = ~mask
        for name in self._NAMES:
            t = getattr(self, name)[keep].contiguous()
            t.grad = torch.zeros_like(t)
            setattr(self, name, t)

    @torch.no_grad()
    def append_points(self, new):
        """scene/gaussian_model.py:533-585 `cat_tensors_to_optimizer` + `densification_postfix` (add_farfield_points): rows are
        appended to every parameter, gradients restart at zero. `new`: dict with the reference's raw attribute names
        (mean, opacity, scale, rotation, rgb, normal, roughness, f0)."""
        src = dict(_xyz="mean", _opacity="opacity", _scaling="scale", _rotation="rotation", _diffuse="rgb", _normal="normal", _roughness="roughness", _f0="f0")
        for name in self._NAMES:
            old = getattr(self, name)
            add = torch.as_tensor(np.asarray(new[src[name]]), dtype=torch.float32, device=old.device).reshape(-1, old.shape[1])
            t = torch.cat([old, add], 0).contiguous()
            t.grad = torch.zeros_like(t)
            setattr(self, name, t)

    # getters read by _export_param_values (gaussian_raytracer.py:41-50); an EditableGaussianModel overrides these
    get_scaling = property(lambda s: torch.exp(s._scaling))
    _get_scaling = property(lambda s: s._scaling)
    _get_rotation = property(lambda s: s._rotation)
    get_xyz = property(lambda s: s._xyz)
    get_diffuse = property(lambda s: s._diffuse)
    get_normal = property(lambda s: s._normal)
    get_roughness = property(lambda s: s._roughness)
    get_f0 = property(lambda s: s._f0)


class GaussianRaytracer:
    OUTPUT_BUFFERS = ("output_rgb", "output_depth", "output_normal", "output_f0", "output_roughness", "output_transmittance",
                      "output_total_transmittance", "output_ray_origin", "output_ray_direction", "output_final")

    EVAL_MODES = ("gather", "full_image")

    def __init__(self, pc, image_width: int, image_height: int, ppll_forward_size=None, ppll_backward_size=None, rank=0, world_size=1,
                 gather_buffers=None, team_help=None, eval_mode="gather", group=None):
        """`eval_mode` (partitioned tracers only; also a per-call argument): what a no-grad render does.
        "gather": every rank traces its own tiles and the images are completed with ONE all-gather - a COLLECTIVE: every rank of
        `group` must make the call (needs a process group whose size equals `world_size`; without one the rank traces the whole image).
        "full_image": this rank traces the whole image itself, no collective - for the usual `if rank == 0: evaluate()` pattern.
        `group`: the process group of the partition's ranks (None = the default group); used by the gradient all-reduce and the gather."""
        assert eval_mode in self.EVAL_MODES, eval_mode
        self.eval_mode, self.group = eval_mode, group
        self.image_width, self.image_height = image_width, image_height
        # partitioned evaluation renders: which framebuffer outputs `gather_outputs` completes on every rank (default: all ten) and
        # whether every no-grad call does it (False: the caller gathers when it needs the images, e.g. once after the 128 accumulated
        # samples of render.py:195-204 - the accumulators of a rank's own pixels live on that rank)
        self.gather_buffers = tuple(gather_buffers) if gather_buffers is not None else self.OUTPUT_BUFFERS
        self.gather_each_render = True
        self._image_gather = None
        kw = {}
        if ppll_forward_size is not None:
            kw["ppll_forward_size"] = int(ppll_forward_size)
        if ppll_backward_size is not None:
            kw["ppll_backward_size"] = int(ppll_backward_size)
        n = pc.get_scaling.shape[0]
        self.cuda_module = make_raytracer(image_width, image_height, n, **kw)
        if self.cuda_module.get_gaussians().mean.shape[0] != n:  # n == 0: the native holder starts with count = 1 (core/gaussians.h:31)
            self.cuda_module.resize(n)
        self.rank, self.world_size = rank, world_size
        self.import_grads = True  # False: a fused host step (trainer.FusedTrainStep) imports the raytracer gradients itself
        if team_help is not None:  # several waves on one heavy tile (egr_set_team_help; library default: on). Only the order of EXACT depth ties of bounce rays depends on it
            self.cuda_module.set_team_help(bool(team_help))
        if world_size > 1:
            self.cuda_module.set_partition(rank, world_size)
            self.cuda_module.use_grad_delta(True)  # grad launches write a per-launch buffer (the first after a fold stores, further ones add): see all_reduce_grads
        config = self.cuda_module.get_config()  # gaussian_raytracer.py:16-25
        config.loss_weight_diffuse.fill_(pc.cfg.loss_weight_diffuse)
        config.loss_weight_specular.fill_(pc.cfg.loss_weight_specular)
        config.loss_weight_normal.fill_(pc.cfg.loss_weight_normal)
        config.loss_weight_depth.fill_(pc.cfg.loss_weight_depth)
        config.loss_weight_f0.fill_(pc.cfg.loss_weight_f0)
        config.loss_weight_roughness.fill_(pc.cfg.loss_weight_roughness)
        config.transmittance_threshold.fill_(pc.cfg.transmittance_threshold)
        config.alpha_threshold.fill_(pc.cfg.alpha_threshold)
        config.exp_power.fill_(pc.cfg.exp_power)
        self.pc = pc
        self._export_param_values()
        self.cuda_module.rebuild_bvh()

    @torch.no_grad()
    def rebuild_bvh(self):  # gaussian_raytracer.py:33-38
        self.cuda_module.resize(self.pc._xyz.shape[0])
        self._export_param_values()
        self.cuda_module.rebuild_bvh()

    @torch.no_grad()
    def _export_param_values(self):  # gaussian_raytracer.py:41-50 (same order)
        g = self.cuda_module.get_gaussians()
        # the eight copy_ calls upstream, as one multi-tensor launch (same order, same semantics)
        torch._foreach_copy_([g.scale, g.rotation, g.mean, g.opacity, g.rgb, g.normal, g.roughness, g.f0],
                             [self.pc._get_scaling, self.pc._get_rotation, self.pc.get_xyz, self.pc._opacity, self.pc.get_diffuse, self.pc.get_normal,
                              self.pc.get_roughness, self.pc.get_f0])

    @torch.no_grad()
    def _import_param_gradients(self):  # gaussian_raytracer.py:53-62
        g = self.cuda_module.get_gaussians()
        # the eight add_ calls upstream, as one multi-tensor launch
        torch._foreach_add_([self.pc._xyz.grad, self.pc._opacity.grad, self.pc._scaling.grad, self.pc._rotation.grad, self.pc._diffuse.grad,
                             self.pc._normal.grad, self.pc._roughness.grad, self.pc._f0.grad],
                            [g.mean.grad, g.opacity.grad, g.scale.grad, g.rotation.grad, g.rgb.grad, g.normal.grad, g.roughness.grad, g.f0.grad])

    def zero_grad(self):  # gaussian_raytracer.py:64-73 (one fill of the flat buffer == the eight zero_() upstream + total_weight kept)
        g = self.cuda_module.get_gaussians()
        n = g.mean.shape[0]
        g.grad_flat[: 21 * n].zero_()  # the eight gradient tensors are the first 21N floats of the flat buffer; total_weight is the tail

    @torch.no_grad()
    def all_reduce_grads(self):
        """Multi-GPU exchange step: the grad launches since the last fold left this rank's gradients + weights in the
        per-launch buffer `grad_delta` ([22N], 88 MB at N=1M; the first launch after a fold stores, further ones add, nobody clears it);
        ONE all-reduce sums it over the ranks, then it is added to the persistent `grad_flat` (whose total_weight tail lives across a
        whole pruning interval) and the library is told that the buffer is consumed. A collective: every rank of the group calls it.
        No-op when the tracer is not partitioned."""
        g = self.cuda_module.get_gaussians()
        if g.grad_delta.numel():
            all_reduce_launch_delta(g.grad_flat, g.grad_delta, self.group, cuda_module=self.cuda_module)  # (reduce + fold + "consumed" in one call: they must not be separated)

    def _partitioned_eval(self, eval_mode=None):
        """Evaluation (no-grad) renders of a partitioned tracer in "gather" mode: with a process group of the partition's size every rank
        traces its own tiles and the images are all-gathered (`gather_outputs`). In "full_image" mode, or without such a group (a single
        process driving one "rank" of a partition, as tools and tests do), the rank traces the whole image itself."""
        import torch.distributed as dist

        if (eval_mode or self.eval_mode) != "gather":
            return False
        return self.world_size > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) == self.world_size

    def _set_full_image(self, full):
        """Whole-image no-grad render on a partitioned tracer (no process group, see _partitioned_eval). The library caches the tile
        order of every partition it has seen, so the flip is a pointer swap: no device sync, no upload."""
        if self.world_size > 1:
            self.cuda_module.set_partition(0 if full else self.rank, 1 if full else self.world_size)

    @torch.no_grad()
    def gather_outputs(self, names=None):
        """Completes the framebuffer outputs `names` (default: self.gather_buffers) of the last partitioned no-grad render on every rank:
        one all-gather of each rank's own pixels, bit-identical to a whole-image render. The per-pixel statistics and
        random_seeds stay per rank. A collective over `self.group`. No-op for an unpartitioned tracer and for one in "full_image" mode."""
        if not self._partitioned_eval():
            return
        fb = self.cuda_module.get_framebuffer()
        if self._image_gather is None:
            self._image_gather = ImageGather(self.image_width, self.image_height, self.rank, self.world_size, fb.output_rgb.device)
        self._image_gather.gather([getattr(fb, n) for n in (names or self.gather_buffers)], self.group)

    @staticmethod
    def blender_rotation(R):
        """gaussian_raytracer.py:95-97: R_c2w_blender = -R; R_c2w_blender[:, 0] *= -1."""
        Rb = -R
        Rb[:, 0] = -Rb[:, 0]
        return Rb

    def __call__(self, viewpoint_camera, target=None, target_diffuse=None, target_specular=None, target_depth=None, target_normal=None,
                 target_roughness=None, target_f0=None, force_update_bvh=False, denoise=False, znear=0.01, zfar=999.9, eval_mode=None):
        assert eval_mode is None or eval_mode in self.EVAL_MODES, eval_mode
        with torch.no_grad():
            R = torch.from_numpy(viewpoint_camera.R).cuda().float() if isinstance(viewpoint_camera.R, np.ndarray) else viewpoint_camera.R.cuda()
            # gaussian_raytracer.py:94-100 (R_c2w_blender = blender_rotation(R); znear / zfar / fov fill_; set_pose(camera_center, R_c2w_blender)) as ONE launch
            self.cuda_module.set_camera(R, viewpoint_camera.camera_center, float(viewpoint_camera.FoVy), float(os.getenv("ZNEAR", znear)), float(os.getenv("ZFAR", zfar)))
            self._export_param_values()
            framebuffer = self.cuda_module.get_framebuffer()
            # CHW -> HWC into framebuffer.target_*, zeros when absent (gaussian_raytracer.py:109-137: six copy_ / zero_ calls) - as ONE launch over the
            # pixels of this rank's own tiles (egr_set_targets_chw; the kernel waits for nothing: a temporary's memory is stream-ordered)
            self.cuda_module.set_targets_chw(target_diffuse, target_specular, target_depth, target_normal, target_roughness, target_f0)
        grads = torch.is_grad_enabled()
        if grads or force_update_bvh:
            # raytrace() follows with the same parameter values: one pass over the cloud writes the snapshot AND the live records
            self.cuda_module.update_bvh(True)
        gathered = not grads and self._partitioned_eval(eval_mode)  # (a collective follows: every rank of the group makes this call - or pass eval_mode="full_image")
        if not grads and not gathered:
            self._set_full_image(True)
        try:
            self.cuda_module.raytrace()
        finally:  # a raytrace that raises (stale BVH, exact-stats gate) must not leave a training rank tracing the whole image
            if not grads and not gathered:
                self._set_full_image(False)
        if gathered and (self.gather_each_render or denoise):
            self.gather_outputs()  # every rank now holds the whole image (the denoiser needs it)
        if denoise:
            self.cuda_module.denoise()
        if grads:
            self.all_reduce_grads()
            if self.import_grads:
                self._import_param_gradients()
        return {"render": framebuffer.output_rgb.clone()}


def render(camera, raytracer: GaussianRaytracer, targets_available=True, force_update_bvh=False, denoise=False, znear=0.01, zfar=999.9, eval_mode=None):
    """gaussian_renderer.py:21-92: returns CHW clones rgb[3,3,H,W], final[1,3,H,W], depth/normal/roughness/f0."""
    do_backprop = torch.is_grad_enabled()
    names = ("original_image", "diffuse_image", "specular_image", "normal_image", "f0_image", "roughness_image", "depth_image")
    tg = {n: (getattr(camera, n, None) if targets_available else None) for n in names}
    with torch.set_grad_enabled(do_backprop):
        raytracer(camera, target=tg["original_image"], target_diffuse=tg["diffuse_image"], target_specular=tg["specular_image"],
                  target_depth=tg["depth_image"], target_normal=tg["normal_image"], target_roughness=tg["roughness_image"],
                  target_f0=tg["f0_image"], force_update_bvh=force_update_bvh, denoise=denoise, znear=znear, zfar=zfar, eval_mode=eval_mode)
    fb = raytracer.cuda_module.get_framebuffer()
    cl = lambda t: t.clone().detach().moveaxis(-1, 1)
    return SimpleNamespace(rgb=cl(fb.output_rgb), final=cl(fb.output_denoised) if denoise else cl(fb.output_final), depth=cl(fb.output_depth),
                           normal=cl(fb.output_normal), roughness=cl(fb.output_roughness), f0=cl(fb.output_f0), target=tg["original_image"],
                           target_diffuse=tg["diffuse_image"], target_specular=tg["specular_image"], target_depth=tg["depth_image"],
                           target_normal=tg["normal_image"], target_roughness=tg["roughness_image"], target_f0=tg["f0_image"])


def camera_from_RT(R, T, FoVy, device="cuda", **images):
    """What scene/cameras.py:22-152 holds for a view, from the dataset's (R, T, FovY) (dataset/blender_dataset.py:62-75: R is the
    camera-to-world rotation in COLMAP axes - "stored transposed" -, T the world-to-camera translation): `R` unchanged, `FoVy`, and
    `camera_center` = the translation of the inverse world-to-view matrix (Camera.update via getWorld2View2) = -R T. Pinned by
    tests/golden/reference_cameras.npz, generated by running the reference's Camera class."""
    R = np.asarray(R, np.float64)
    centre = -R @ np.asarray(T, np.float64)
    cam = SimpleNamespace(R=R.astype(np.float32), T=np.asarray(T, np.float32), FoVy=float(FoVy),
                          camera_center=torch.as_tensor(centre.astype(np.float32), device=device))
    for k, v in images.items():
        setattr(cam, k, v)
    return cam


def camera_from_c2w(origin, c2w_blender, fov, **images):
    """Builds a duck-typed reference `Camera` (scene/cameras.py:22) whose .R reproduces `c2w_blender` after the
    caller-side conversion above (R = blender rotation with the inverse flip applied)."""
    Rb = torch.as_tensor(np.asarray(c2w_blender), dtype=torch.float32).clone()
    R = Rb.clone()
    R[:, 0] = -R[:, 0]
    R = -R
    cam = SimpleNamespace(R=R.cuda(), FoVy=float(fov), camera_center=torch.as_tensor(np.asarray(origin), dtype=torch.float32).cuda())
    for k, v in images.items():
        setattr(cam, k, v)
    return cam
