"""Worker of tests/test_hip_multirank.py::test_exchange_step_through_rccl_at_world_size_1: the exchange step of the multi-GPU path
through RCCL itself (backend "nccl") on the one GPU of the test box. A one-rank group sums nothing, but everything else of the N > 1
path runs for real: ncclCommInitRank, the launch storing into the per-launch buffer (grad_delta, egr_set_grad_overwrite), the
all-reduce ON THE DEVICE BUFFER enqueued behind the launch on torch's stream, the fold into the persistent gradients, the gradient
import - three training iterations compared with a plain tracer - and the evaluation render's all-gather. Prints RCCL_OK."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("editable-gaussian-reflections_amd.synthetic")
ren = importlib.import_module("editable-gaussian-reflections_amd.renderer")
par = importlib.import_module("editable-gaussian-reflections_amd.parallel")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
par.REDUCE_AT_WORLD_1 = True
W, H, N = 320, 192, 20000
g = syn.make_scene(N, "trained", seed=5)
cam = syn.default_camera()
tg = syn.make_targets(W, H)
images = {k + "_image": torch.tensor(v).cuda().moveaxis(-1, 0).contiguous() for k, v in tg.items()}
camera = ren.camera_from_c2w(cam["origin"], cam["c2w"], cam["fov"], **images)
kw = dict(ppll_forward_size=40_000_000, ppll_backward_size=40_000_000)
delta = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, **kw)
delta.cuda_module.use_grad_delta(True)  # the N > 1 gradient path on a one-rank group
plain = ren.GaussianRaytracer(ren.GaussianParams(g), W, H, **kw)
gd, gp = delta.cuda_module.get_gaussians(), plain.cuda_module.get_gaussians()
assert gd.grad_delta.numel() == 22 * N and gd.grad_delta.is_cuda
gd.grad_delta.fill_(123.0)  # stale content: the launch stores, nobody clears
for it in range(3):
    for rt in (delta, plain):
        rt.zero_grad()
        rt.cuda_module.get_metadata().total_num_calls.fill_(it)
        ren.render(camera, rt)  # no host synchronisation between the launch and the collective: stream order must do
    torch.cuda.synchronize()
    a, b = gd.grad_flat, gp.grad_flat
    err = float((a[: 21 * N] - b[: 21 * N]).abs().max()) / float(b[: 21 * N].abs().max())
    werr = float((a[21 * N:] - b[21 * N:]).abs().max()) / float(b[21 * N:].abs().max())
    perr = float((delta.pc._xyz.grad - plain.pc._xyz.grad).abs().max()) / float(plain.pc._xyz.grad.abs().max())
    assert err < 1e-5 and werr < 1e-5 and perr < 1e-5, (it, err, werr, perr)
    for rt in (delta, plain):
        for p in rt.pc.parameters():
            p.grad.zero_()
assert delta.cuda_module.get_counters()[11] == 0
# the evaluation render's all-gather (parallel.ImageGather) through RCCL: one rank owns every pixel, the buffers must come back unchanged
with torch.no_grad():
    plain(camera)
fb = plain.cuda_module.get_framebuffer()
bufs = [getattr(fb, n) for n in ren.GaussianRaytracer.OUTPUT_BUFFERS]
keep = [b.clone() for b in bufs]
par.ImageGather(W, H, 0, 1, bufs[0].device).gather(bufs)
torch.cuda.synchronize()
assert all(torch.equal(a, b) for a, b in zip(bufs, keep))
print("RCCL_OK", err, werr, flush=True)
dist.destroy_process_group()
