"""Worker of tests/test_hip_multirank.py: one of two ranks that SHARE cuda:0 (gloo rendezvous; RCCL refuses duplicate devices).
Every rank traces its tiles with the HIP path, the product's exchange step (renderer.GaussianRaytracer.all_reduce_grads ->
parallel.all_reduce_launch_delta) sums the per-launch buffers, and the result is compared with an unpartitioned tracer in the same
process. Prints MULTIRANK_OK on rank 0."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")

os.environ.setdefault("EGR_RAYS_PER_TASK", "64")  # same task shape for the partitioned and the unpartitioned tracer (this image is tiny: a partitioned rank
                                                   # would otherwise switch to 8x4-pixel tasks, whose candidate lists - and exactly tied hits - come in another order)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
W, H, N = 160, 96, 5000
g = syn.make_scene(N, "trained", seed=5)
cam = syn.default_camera()
tg = syn.make_targets(W, H)
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
kw = dict(ppll_forward_size=20_000_000, ppll_backward_size=20_000_000)
part = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, rank=rank, world_size=world, **kw)
full = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, **kw)
for rt in (part, full):
    rt.cuda_module.get_config().jitter_primary_rays.fill_(False)
gp, gf = part.cuda_module.get_gaussians(), full.cuda_module.get_gaussians()
assert gp.grad_delta.numel() == 22 * N and gf.grad_delta.numel() == 0
for it in range(3):  # three training iterations; total_weight is only cleared at a prune (train.py:238-245)
    for rt in (part, full):
        rt.zero_grad()
        rt.cuda_module.get_metadata().total_num_calls.fill_(it)
        ren.render(camera, rt)
    torch.cuda.synchronize()
    a, b = gp.grad_flat, gf.grad_flat
    scale = float(b[: 21 * N].abs().max())
    err = float((a[: 21 * N] - b[: 21 * N]).abs().max()) / scale
    werr = float((a[21 * N:] - b[21 * N:]).abs().max()) / float(b[21 * N:].abs().max())
    assert err < 1e-5 and werr < 1e-5, (it, err, werr)  # summed over ranks == single rank, weights grow linearly (not x world per iteration)
    # the per-launch buffer holds the SUM over the ranks of this launch (stored by the launch, reduced in place, never cleared)
    derr = float((gp.grad_delta[: 21 * N] - b[: 21 * N]).abs().max()) / scale
    assert derr < 1e-5, (it, derr)
    # the model-side gradients (python import) agree too
    perr = float((part.pc._xyz.grad - full.pc._xyz.grad).abs().max()) / float(full.pc._xyz.grad.abs().max())
    assert perr < 1e-5, perr
    for rt in (part, full):
        for p in rt.pc.parameters():
            p.grad.zero_()
cp, cf = part.cuda_module.get_counters(), full.cuda_module.get_counters()
rays = torch.tensor([float(cp[0])], dtype=torch.float64)
dist.all_reduce(rays)
assert int(rays.item()) == cf[0] == W * H and cp[11] == 0 and cf[11] == 0
# evaluation renders of a partitioned tracer: every rank traces ITS tiles, one all-gather completes the images on every rank -
# bit for bit the whole-image render of the unpartitioned tracer, all ten output buffers
def same_images(tag):
    fp, ff = part.cuda_module.get_framebuffer(), full.cuda_module.get_framebuffer()
    for name in ren.GaussianRaytracer.OUTPUT_BUFFERS:
        assert torch.equal(getattr(fp, name), getattr(ff, name)), (tag, name)
with torch.no_grad():
    for rt in (part, full):
        rt.cuda_module.get_metadata().total_num_calls.zero_()
        rt(camera)
same_images("one sample")
rays = torch.tensor([float(part.cuda_module.get_counters()[0])], dtype=torch.float64)
dist.all_reduce(rays)
assert int(rays.item()) == W * H and part.cuda_module.get_counters()[0] < W * H  # nobody traced the whole image
# the usual `if rank == 0: evaluate()` pattern: eval_mode="full_image" is NOT a collective - rank 0 alone traces the whole image (rank 1 makes no call)
if rank == 0:
    with torch.no_grad():
        for rt in (part, full):
            rt.cuda_module.get_metadata().total_num_calls.zero_()
        part(camera, eval_mode="full_image")
        full(camera)
    same_images("rank 0 alone, full_image")
    assert part.cuda_module.get_counters()[0] == W * H
dist.barrier()
# render.py:195-209: accumulated samples (jitter on), the gather deferred to the end of the loop, then the denoiser on the whole image
for rt in (part, full):
    rt.cuda_module.get_config().jitter_primary_rays.fill_(True)
    rt.cuda_module.get_config().accumulate_samples.fill_(True)
    rt.cuda_module.reset_accumulators()
    rt.cuda_module.get_metadata().total_num_calls.zero_()
part.gather_each_render = False
with torch.no_grad():
    for _ in range(4):
        for rt in (part, full):
            ren.render(camera, rt)
part.gather_outputs()
same_images("four accumulated samples, one gather")
part.gather_each_render = True
with torch.no_grad():
    for rt in (part, full):
        pk = ren.render(camera, rt, denoise=True)
same_images("fifth sample")
assert torch.equal(part.cuda_module.get_framebuffer().output_denoised, full.cuda_module.get_framebuffer().output_denoised)
for rt in (part, full):
    rt.cuda_module.get_config().jitter_primary_rays.fill_(False)
    rt.cuda_module.get_config().accumulate_samples.fill_(False)
part.zero_grad()
ren.render(camera, part)  # and the next training iteration is partitioned again
assert part.cuda_module.get_counters()[0] == cp[0]
dist.barrier()
if rank == 0:
    print("MULTIRANK_OK", err, werr, flush=True)
dist.destroy_process_group()
