"""Multi-GPU decomposition of the hot path (nothing like it exists in the reference).

The path shards over PIXELS: forward is independent per pixel, backward contributions sum over pixels. One
process per GPU holds the full Gaussian set + BVH and traces the 16x16-pixel macro tiles whose index is congruent
to its rank (round-robin = load balance between empty and dense image regions); the only exchange step is ONE
all-reduce (sum) of the contiguous [22N] per-launch gradient buffer per iteration (RCCL over xGMI on the GPU box, gloo in the
CPU tests). After it every rank holds identical gradients, so the replicated optimiser steps stay in lock-step.
"""
import numpy as np

MACRO_TILE = 16  # must match EGR_MACRO_TILE in csrc/egr_internal.hpp
WAVE_TILE = 8


def macro_tiles(width, height):
    return (width + MACRO_TILE - 1) // MACRO_TILE, (height + MACRO_TILE - 1) // MACRO_TILE


def num_tasks_for_rank(width, height, rank, world):
    """Python mirror of egr_num_tasks_for_rank (csrc/trace.hip): wave tiles (8x8) owned by `rank`."""
    mtx, mty = macro_tiles(width, height)
    M = mtx * mty
    if rank >= M:
        return 0
    return 4 * ((M - rank + world - 1) // world)


def owner_map(width, height, world):
    """[H,W] int array: which rank traces each pixel (mirror of task_geom in csrc/egr_state.hpp)."""
    mtx, _ = macro_tiles(width, height)
    yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    return ((yy // MACRO_TILE) * mtx + (xx // MACRO_TILE)) % world


def all_reduce_flat(flat, group=None):
    """Sum a flat gradient buffer over ranks in place (one collective). No-op without an initialised process group.
    Backend "nccl" is RCCL over xGMI on the GPU box. With "gloo" (CPU tests; two ranks sharing ONE GPU in the -m gpu suite, where
    RCCL refuses duplicate devices) a device buffer is staged through the host."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def all_reduce_launch_delta(grad_flat, grad_delta, group=None):
    """The exchange step of one training iteration: `grad_delta` holds what THIS launch added on this rank (the kernels
    accumulate into it instead of `grad_flat`, csrc/torch_binding.cpp: GaussianDataHolder::grad_delta). Sum it over the
    ranks with one collective, fold it into the persistent buffer, empty it for the next launch. Reducing `grad_flat`
    itself would multiply everything it already holds - total_weight since the last prune, gradients of an earlier
    launch that were not zeroed - by the world size each time."""
    all_reduce_flat(grad_delta, group)
    grad_flat.add_(grad_delta)
    grad_delta.zero_()
    return grad_flat


GRAD_LAYOUT = [("dL_drgb", 3), ("dL_dnormal", 3), ("dL_df0", 3), ("dL_droughness", 1), ("dL_dopacity", 1), ("dL_dscale", 3),
               ("dL_dmean", 3), ("dL_drotation", 4), ("total_weight", 1)]  # tensor-major order inside grad_flat


def split_flat(flat, n):
    """Views of the [22N] buffer with the reference's per-tensor shapes (core/gaussians.h:15-24)."""
    out, off = {}, 0
    for name, c in GRAD_LAYOUT:
        out[name] = flat[off:off + n * c].reshape(n, c)
        off += n * c
    return out
