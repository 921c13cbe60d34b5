"""Worker of tests/test_hip_sequences.py::test_config4_substitute_two_ranks_at_size - BASELINE config 4 ("full training loop, 1080p,
image-tile split + gradient all-reduce") with what this box has: the SYNTHETIC room instead of shiny_kitchen (the dataset is not
here) and two ranks that SHARE cuda:0 over gloo instead of 8 GPUs over RCCL (the driver owns multi-GPU runs).

The loop is train.py:211-263 with the fused host step: per iteration another camera, render() in grad mode (export, refit, forward +
backward, all-reduce of the launch's [22N] buffer), scale decay + Adam + clamps + zero_grads; at the pruning interval
`prune_points(total_weight / interval < min_weight)`, `total_weight.zero_()`, `rebuild_bvh()` BEFORE the optimizer step
(train.py:238-247); in the middle of an interval far-field points are appended and the tracer resized (train.py:256-260).
Every rank also runs the same loop on an UNPARTITIONED tracer and compares, iteration by iteration: gradients, total_weight,
the prune mask, the parameters - and on a SECOND unpartitioned tracer (`twin`), because a training loop is not reproducible to the
last bit even on one GPU: float atomics add a gaussian's contributions in varying order, and Adam with eps = 1e-15 (gaussian_model.py:338)
turns a gradient element whose sign depends on that order - a sum that cancels to ~0 - into an update of +-lr; from the second
iteration on a few dozen of the 6.6M parameter elements differ by up to 2 lr between ANY two runs, and the gaussians they belong to
render slightly differently. The bar for the partitioned run is therefore: bit-level agreement where that is defined (iteration 1,
the prune masks), and no further from the single-rank run than the single-rank run is from its own twin. Prints CONFIG4_OK on rank 0."""
import importlib
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "editable-gaussian-reflections_amd"
syn = importlib.import_module(PKG + ".synthetic")
ren = importlib.import_module(PKG + ".renderer")
tr = importlib.import_module(PKG + ".trainer")
knn = importlib.import_module(PKG + ".simple_knn")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
W, H = int(os.environ.get("C4_W", 1920)), int(os.environ.get("C4_H", 1080))
N, ITERS, INTERVAL, FARFIELD_AT, K_FAR = int(os.environ.get("C4_N", 300_000)), int(os.environ.get("C4_ITERS", 9)), 4, 6, 20_000
MIN_WEIGHT = 0.1  # config.py:32
LRS = dict(xyz=0.00016, normal=0.0025, roughness=0.0025, f0=0.0025, f_dc=0.005, opacity=0.025, scaling=0.005, rotation=0.001)  # config.py:62-72

g0 = syn.make_scene(N, "trained", seed=21)
tg = syn.make_targets(W, H)
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
eyes = [(-1.7, -1.2, 0.4), (-1.5, 1.1, 0.2), (-0.4, -1.6, 0.6), (-1.8, 0.0, -0.3), (-1.0, -1.0, 0.9), (-1.6, 0.6, 0.5), (-0.8, 1.4, 0.1), (-1.3, -0.2, 0.7), (-1.7, -0.9, -0.5)]
cams = [ren.camera_from_c2w(np.asarray(e, np.float32), syn.look_at(e, (1.2, 0.4 - 0.2 * i, -0.7)).astype(np.float32), 0.6911, **images) for i, e in enumerate(eyes)]


class Run:
    def __init__(self, partitioned):
        self.pc = ren.GaussianParams(g0)
        kw = dict(rank=rank, world_size=world) if partitioned else {}
        self.rt = ren.GaussianRaytracer(self.pc, W, H, ppll_forward_size=200_000_000, ppll_backward_size=150_000_000, **kw)
        self.m = self.rt.cuda_module
        self.step = tr.FusedTrainStep(self.pc, self.rt, LRS, scale_decay=0.9999,
                                      xyz_schedule=dict(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=32_000))
        self.m.get_metadata().total_num_calls.zero_()


part, full, twin = Run(True), Run(False), Run(False)
assert part.m.get_gaussians().grad_delta.numel() == 22 * N and full.m.get_gaussians().grad_delta.numel() == 0


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def outliers(a, b, tol=1e-5):
    """elements further apart than tol * max|b| (Adam with eps = 1e-15 turns a gradient element whose SIGN depends on the order of the
    float atomics - a sum that cancels to ~0 - into an update of +-lr; the reference has the same run-to-run behaviour)"""
    return int(((a - b).abs() > tol * float(b.abs().max())).sum())


log = []
for it in range(1, ITERS + 1):
    cam = cams[(it - 1) % len(cams)]
    for r in (part, full, twin):
        r.step.update_learning_rate(it)
        ren.render(cam, r.rt)
    torch.cuda.synchronize()
    gp, gf, gt = part.m.get_gaussians(), full.m.get_gaussians(), twin.m.get_gaussians()
    n = gf.mean.shape[0]
    assert gp.mean.shape[0] == n and gt.mean.shape[0] == n
    e_grad = rel(gp.grad_flat[: 21 * n], gf.grad_flat[: 21 * n])
    e_w = rel(gp.grad_flat[21 * n:], gf.grad_flat[21 * n:])
    w_out = outliers(gp.total_weight, gf.total_weight)
    e_grad_twin, w_out_twin = rel(gt.grad_flat[: 21 * n], gf.grad_flat[: 21 * n]), outliers(gt.total_weight, gf.total_weight)
    g_out, g_out_twin = outliers(gp.grad_flat[: 21 * n], gf.grad_flat[: 21 * n]), outliers(gt.grad_flat[: 21 * n], gf.grad_flat[: 21 * n])
    assert part.m.get_counters()[11] == 0 and full.m.get_counters()[11] == 0, (it, part.m.get_counters()[11], full.m.get_counters()[11])
    pruned = 0
    if it % INTERVAL == 0:  # train.py:238-245
        wf = gf.total_weight / INTERVAL
        t = torch.tensor([MIN_WEIGHT], dtype=torch.float64)
        while int(((wf - float(t)).abs() < 1e-4 * float(t)).sum()) > 0:  # a threshold no gaussian sits on (float-atomic noise is ~1e-6 of a weight)
            t *= 1.001
        dist.broadcast(t, 0)
        masks = [(r.m.get_gaussians().total_weight / INTERVAL < float(t)).squeeze(1) for r in (part, full, twin)]
        # by now the three runs hold slightly different parameters (see the header), so a gaussian whose weight sits within that noise of
        # the threshold may fall on either side: the partitioned run may disagree with the single-rank run on no more gaussians than
        # its twin does (+ a handful); all three then prune the single-rank run's mask so that they stay comparable row by row
        mism, mism_twin = int((masks[0] != masks[1]).sum()), int((masks[2] != masks[1]).sum())
        assert mism <= 3 * mism_twin + 5, (it, mism, mism_twin)
        # (a real multi-GPU run prunes by `part`'s mask, identical on every rank because the all-reduced total_weight is; the reference
        # run of THIS test lives once per process and each copy carries its own noise, so rank 0's mask is the one everybody applies)
        pm = [masks[0].cpu().clone() for _ in range(world)]
        dist.all_gather(pm, masks[0].cpu())
        assert all(torch.equal(pm[0], x) for x in pm), "the partitioned runs of the two ranks disagree on the prune mask"
        ref_mask = masks[1].cpu()
        dist.broadcast(ref_mask, 0)
        masks[1] = ref_mask.cuda()
        pruned = int(masks[1].sum())
        assert 0 < pruned < n // 2, pruned
        for r, mask in zip((part, full, twin), (masks[1], masks[1], masks[1])):
            r.pc.prune_points(mask)
            r.step.prune(~mask)
            r.m.get_gaussians().total_weight.zero_()
            r.rt.rebuild_bvh()
            assert r.m.get_gaussians().mean.shape[0] == n - pruned and r.m.check_bvh() == 0
    for r in (part, full, twin):
        r.step.step()
    if it == FARFIELD_AT:  # train.py:256-260 / gaussian_model.py:233-283, mid-interval: total_weight of the existing rows must survive the resize
        rng = np.random.default_rng(5)
        xyz = (np.clip(rng.normal(size=(K_FAR, 3)), -3, 3) * 4.0).astype(np.float32)
        xyz = xyz[np.abs(xyz).max(1) > 2.5][: K_FAR // 2]  # outside the room
        d2 = knn.distCUDA2(torch.from_numpy(xyz).cuda()).clamp_min(1e-7)
        new = dict(mean=xyz, scale=torch.log(torch.sqrt(d2) * 0.1)[:, None].repeat(1, 3).cpu().numpy(), rotation=np.tile(np.array([[1, 0, 0, 0]], np.float32), (len(xyz), 1)),
                   opacity=np.full((len(xyz), 1), math.log(0.1 / 0.9), np.float32), rgb=np.full((len(xyz), 3), 0.2, np.float32), normal=np.zeros((len(xyz), 3), np.float32),
                   f0=np.full((len(xyz), 3), 0.04, np.float32), roughness=np.zeros((len(xyz), 1), np.float32))
        for r in (part, full, twin):
            before = r.m.get_gaussians().total_weight.clone()
            r.pc.append_points(new)
            r.step.extend(len(xyz))
            r.rt.rebuild_bvh()
            tw = r.m.get_gaussians().total_weight
            assert tw.shape[0] == before.shape[0] + len(xyz) and torch.equal(tw[: before.shape[0]], before) and float(tw[before.shape[0]:].abs().max()) == 0.0
    torch.cuda.synchronize()
    n2 = full.pc._xyz.shape[0]
    assert part.pc._xyz.shape[0] == n2 and twin.pc._xyz.shape[0] == n2
    p_out = sum(outliers(a, b) for a, b in zip(part.pc.parameters(), full.pc.parameters()))
    p_out_twin = sum(outliers(a, b) for a, b in zip(twin.pc.parameters(), full.pc.parameters()))
    p_rel = max(rel(a, b) for a, b in zip(part.pc.parameters(), full.pc.parameters()))
    log.append(dict(it=it, n=n, mask_mismatch=(mism, mism_twin) if it % INTERVAL == 0 else None, grad=e_grad, grad_twin=e_grad_twin, grad_outliers=g_out, grad_outliers_twin=g_out_twin, weight=e_w, weight_outliers=w_out, weight_outliers_twin=w_out_twin, pruned=pruned,
                    param_outliers=p_out, param_outliers_twin=p_out_twin, param_worst=p_rel))
    if rank == 0:
        print("CONFIG4", log[-1], flush=True)
    if it == 1:  # identical parameters on all three runs: the partition + all-reduce reproduces the single-rank launch to float-atomic reordering
        assert e_grad < 1e-5 and e_w < 1e-5 and w_out == 0, log[-1]
    # later: as close to the single-rank run as its own twin is - counted in elements further apart than 1e-5 of the tensor maximum
    # (an order of magnitude + a floor: the counts are tens to hundreds out of 6.3M elements and heavy-tailed - one differing large
    # gaussian touches many neighbours; the maximum itself is one outlier's size and fluctuates between 1e-5 and 1e-2)
    assert g_out <= 10 * g_out_twin + 200 and w_out <= 10 * w_out_twin + 100 and p_out <= 10 * p_out_twin + 200 and e_grad < 2e-2, log[-1]
    lr_max = max(LRS.values())
    p_abs = max(float((a - b).abs().max()) for a, b in zip(part.pc.parameters(), full.pc.parameters()))
    assert p_abs <= 4.0 * lr_max * it + 1e-6, (p_abs, log[-1])  # nothing beyond what Adam can move a parameter in `it` steps

with torch.no_grad():  # evaluation render of the partitioned tracer: whole image on every rank
    for r in (part, full, twin):
        r.m.get_metadata().total_num_calls.zero_()
        r.rt(cams[0])
ip, iff, itw = part.m.get_framebuffer().output_final, full.m.get_framebuffer().output_final, twin.m.get_framebuffer().output_final
mse, mse_twin = float(((ip - iff) ** 2).mean()), float(((itw - iff) ** 2).mean())
# (after nine iterations of run-to-run divergence a handful of gaussians differ visibly in ANY two runs; the twin's distance is printed
# next to the partitioned run's, the bar is an absolute 30 dB: heavy-tailed, a ratio of two such numbers means nothing)
if rank == 0:
    print("CONFIG4 evaluation render: mse partitioned vs single-rank", mse, "twin vs single-rank", mse_twin, flush=True)
rays_part = torch.tensor([float(part.m.get_counters()[0])], dtype=torch.float64)
dist.all_reduce(rays_part)  # every rank traced its own tiles, the images were all-gathered (renderer.gather_outputs)
assert int(rays_part.item()) == W * H and part.m.get_counters()[0] < W * H and mse < 1e-3, (mse, mse_twin)
dist.barrier()
if rank == 0:
    print("CONFIG4_OK", log[-1], flush=True)
dist.destroy_process_group()
