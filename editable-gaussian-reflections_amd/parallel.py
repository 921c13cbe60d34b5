"""Multi-GPU decomposition of the hot path (nothing like it exists in the reference).

The path shards over PIXELS: forward is independent per pixel, backward contributions sum over pixels. One
process per GPU holds the full Gaussian set + BVH and traces its share of the 16x16-pixel macro tiles: the tiles are dealt
round-robin along a Z-curve over the image (`tile_owner`: every rank owns one tile of every compact block of `world` tiles = load
balance between empty and dense image regions in both directions); the only exchange step is ONE
all-reduce (sum) of the contiguous [22N] per-launch gradient buffer per iteration (RCCL over xGMI on the GPU box, gloo in the
CPU tests). After it every rank holds identical gradients, so the replicated optimiser steps stay in lock-step.
"""
import numpy as np

MACRO_TILE = 16  # must match EGR_MACRO_TILE in csrc/egr_internal.hpp
REDUCE_AT_WORLD_1 = False  # True: the collectives also run in a one-rank group (tests/rccl_worker.py: RCCL initialisation, the all-reduce on the
                           # device buffer and its stream ordering against the launch are then exercised on a one-GPU box)
WAVE_TILE = 8


def macro_tiles(width, height):
    return (width + MACRO_TILE - 1) // MACRO_TILE, (height + MACRO_TILE - 1) // MACRO_TILE


def _part1by1(x):
    x = x.astype(np.uint32) & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF
    x = (x | (x << 4)) & 0x0F0F0F0F
    x = (x | (x << 2)) & 0x33333333
    x = (x | (x << 1)) & 0x55555555
    return x


def tile_owner(width, height, world):
    """[mty, mtx] int array: the rank that owns each 16x16 macro tile. Python mirror of egr_build_task_order (csrc/trace.hip): the macro
    tiles are sorted along a Z-curve over (mx, my); the tile at position i of that order belongs to rank (i + i // world) % world -
    every run of `world` positions (a compact 2-D block of the image) holds each rank once, rotated by the block index."""
    mtx, mty = macro_tiles(width, height)
    my, mx = np.meshgrid(np.arange(mty), np.arange(mtx), indexing="ij")
    key = (_part1by1(mx) | (_part1by1(my) << 1)).reshape(-1)
    order = np.argsort(key, kind="stable")  # (keys are distinct)
    pos = np.empty(mtx * mty, np.int64)
    pos[order] = np.arange(mtx * mty)
    return ((pos + pos // world) % world).reshape(mty, mtx)


def num_tasks_for_rank(width, height, rank, world):
    """Python mirror of egr_num_tasks_for_rank (csrc/trace.hip): wave tiles (8x8) owned by `rank`."""
    return 4 * int((tile_owner(width, height, world) == rank).sum())


def owner_map(width, height, world):
    """[H,W] int array: which rank traces each pixel (mirror of task_geom in csrc/egr_state.hpp)."""
    own = tile_owner(width, height, world)
    yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    return own[yy // MACRO_TILE, xx // MACRO_TILE]


def rank_of_tile(width, height, world, tile_index):
    """The rank whose partition is exactly / contains macro tile `tile_index` (= my * mtx + mx). With world = number of macro tiles
    every rank owns one tile: tests use that to trace single tiles through the product's own partition."""
    return int(tile_owner(width, height, world).reshape(-1)[tile_index])


def all_reduce_flat(flat, group=None):
    """Sum a flat gradient buffer over ranks in place (one collective). No-op without an initialised process group.
    Backend "nccl" is RCCL over xGMI on the GPU box. With "gloo" (CPU tests; two ranks sharing ONE GPU in the -m gpu suite, where
    RCCL refuses duplicate devices) a device buffer is staged through the host."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or REDUCE_AT_WORLD_1):
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def all_reduce_launch_delta(grad_flat, grad_delta, group=None, cuda_module=None):
    """The exchange step of one training iteration. `grad_delta` holds what the grad launches SINCE THE LAST FOLD produced on this rank: the
    first launch after `grad_delta_consumed()` STORES its sums there (rows without a contribution store zeros: nobody clears the buffer),
    every further launch before the next fold ADDS (csrc/torch_binding.cpp GaussianDataHolder::grad_delta, egr_set_grad_overwrite,
    egr_grad_delta_consumed). This sums the buffer over the ranks with one collective and folds it into the persistent `grad_flat`:
    two passes over [22N] per iteration besides the collective. Reducing `grad_flat` itself would multiply everything it already
    holds - total_weight since the last prune, gradients of an earlier launch that were not zeroed - by the world size each time.

    THE LIBRARY MUST THEN BE TOLD that the buffer is consumed, or the next launch adds to the already rank-summed values and the
    next reduce sums them again: pass `cuda_module` (the Raytracer) and this function does it; a caller that passes none calls
    `cuda_module.grad_delta_consumed()` itself right after (GaussianRaytracer.all_reduce_grads does)."""
    all_reduce_flat(grad_delta, group)
    grad_flat.add_(grad_delta)
    if cuda_module is not None:
        cuda_module.grad_delta_consumed()
    return grad_flat


class ImageGather:
    """Evaluation renders of a partitioned tracer ("framebuffer outputs are all-gathered only for evaluation renders").
    Under no_grad every rank traces ITS OWN tiles; `gather` then completes the listed [S,H,W,C] buffers on every rank with ONE
    all-gather of the packed pixels of each rank (copies only: the assembled image equals a whole-image render bit for bit).
    Per frame and rank at 1080p, all ten output buffers: 62 MB sent, 436 MB received at world 8 (the six buffers render() returns:
    37 / 261 MB)."""

    def __init__(self, width, height, rank, world, device):
        import torch

        own = owner_map(width, height, world).reshape(-1)
        self.rank, self.world = rank, world
        idx = [np.flatnonzero(own == r) for r in range(world)]
        self.count = [len(i) for i in idx]
        self.pad = max(self.count)
        self.idx = [torch.as_tensor(i, dtype=torch.long, device=device) for i in idx]

    def gather(self, buffers, group=None):
        import torch
        import torch.distributed as dist

        views = [b.view(b.shape[0], -1, b.shape[-1]) for b in buffers]  # [S, P, C]
        mine = self.idx[self.rank]
        parts = [v[:, mine, :].reshape(-1) for v in views]
        per_pixel = sum(v.shape[0] * v.shape[2] for v in views)
        send = torch.zeros(self.pad * per_pixel, dtype=buffers[0].dtype, device=buffers[0].device)
        send[: self.count[self.rank] * per_pixel] = torch.cat(parts)
        host = send.is_cuda and dist.get_backend(group) == "gloo"  # (two test ranks sharing one GPU: staged through the host)
        if host:
            send = send.cpu()
        recv = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(recv, send, group=group)
        for r in range(self.world):
            if r == self.rank:
                continue
            chunk, off, n = recv[r].to(buffers[0].device) if host else recv[r], 0, self.count[r]
            for v in views:
                sz = v.shape[0] * n * v.shape[2]
                v[:, self.idx[r], :] = chunk[off:off + sz].view(v.shape[0], n, v.shape[2])
                off += sz
        return buffers


GRAD_LAYOUT = [("dL_drgb", 3), ("dL_dnormal", 3), ("dL_df0", 3), ("dL_droughness", 1), ("dL_dopacity", 1), ("dL_dscale", 3),
               ("dL_dmean", 3), ("dL_drotation", 4), ("total_weight", 1)]  # tensor-major order inside grad_flat


def split_flat(flat, n):
    """Views of the [22N] buffer with the reference's per-tensor shapes (core/gaussians.h:15-24)."""
    out, off = {}, 0
    for name, c in GRAD_LAYOUT:
        out[name] = flat[off:off + n * c].reshape(n, c)
        off += n * c
    return out
