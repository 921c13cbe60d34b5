// Per-ray state layout + task geometry (trace.hip; the step epilogue in egr_epilogue.hpp is the contraction-free part).
#pragma once
#include "egr_device.hpp"
#include "egr_internal.hpp"

namespace {
// ---- internal per-ray state (floats, SoA: state[field * stride + task*64 + lane]) -------------------------
enum : int {
    F_RAY_O = 0, F_RAY_D = 3, F_SEED = 6, F_STEPS = 7, F_ALIVE = 8, F_STEP_BASE = 9,
    S_RGB = 0, S_DEPTH = 3, S_NORMAL = 4, S_F0 = 7, S_ROUGH = 10, S_T = 11, S_TTOT = 12,
    S_REM_RGB = 13, S_REM_DEPTH = 16, S_REM_NORMAL = 17, S_REM_F0 = 20, S_REM_ROUGH = 23,
    S_THR = 24, S_NEXT_O = 27, S_NEXT_D = 30, S_NHITS = 33, S_COUNT = 34,
    F_TOTAL = F_STEP_BASE + EGR_NSTEPS * S_COUNT
};
#define SF(step, f) (F_STEP_BASE + (step) * S_COUNT + (f))

struct StateRef {
    float *base;
    uint32_t stride, idx;
    EGR_DI float ld(int f) const { return base[(size_t)f * stride + idx]; }
    EGR_DI void st(int f, float x) const { base[(size_t)f * stride + idx] = x; }
    EGR_DI f3 ld3(int f) const { return mk3(ld(f), ld(f + 1), ld(f + 2)); }
    EGR_DI void st3(int f, f3 x) const { st(f, x.x), st(f + 1, x.y), st(f + 2, x.z); }
};

struct TaskGeom {
    int px, py;
    uint32_t pixel_id;
    bool inside;
};
// task -> pixels. Tasks enumerate the 16x16 macro tiles owned by this rank (every world-th tile) in the order of v.task_macro:
// the task range is cut into 8 contiguous chunks (one per XCD queue), each chunk is a compact 2-D block of the image walked
// along a Z-curve, so the waves resident on one XCD at any time cover a compact patch (their rays meet the same BVH nodes /
// records -> per-XCD L2 hits, not fabric traffic). A macro tile holds 4 tasks of 8x8 pixels (one lane per ray, the default),
// or 8 of 8x4 / 16 of 4x4 with the upper lanes idle in the per-ray phases: a rank of a multi-GPU partition owns about as many 8x8
// tiles as there are wave slots, so its launch lasts as long as its heaviest tile - smaller tasks cut that tail, and the pair
// walk keeps all 64 lanes busy whatever the number of rays (trace.hip: egr_make_view chooses).
EGR_DI TaskGeom task_geom(const DeviceView &v, uint32_t task, int lane) {
    const uint32_t mtx = (uint32_t)(v.width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    const uint32_t m = v.task_macro[task >> v.task_shift]; // this rank's macro tiles in cache-friendly order (see egr_build_task_order)
    const uint32_t sub = task & ((1u << v.task_shift) - 1u);
    const int x0 = (int)(m % mtx) * EGR_MACRO_TILE, y0 = (int)(m / mtx) * EGR_MACRO_TILE;
    TaskGeom g;
    if (v.task_shift == 2u) { // 2 x 2 tasks of 8 x 8
        g.px = x0 + (int)(sub & 1u) * 8 + (lane & 7), g.py = y0 + (int)(sub >> 1) * 8 + (lane >> 3);
    } else if (v.task_shift == 3u) { // 2 x 4 tasks of 8 x 4
        g.px = x0 + (int)(sub & 1u) * 8 + (lane & 7), g.py = y0 + (int)(sub >> 1) * 4 + ((lane >> 3) & 3);
    } else { // 4 x 4 tasks of 4 x 4
        g.px = x0 + (int)(sub & 3u) * 4 + (lane & 3), g.py = y0 + (int)(sub >> 2) * 4 + ((lane >> 2) & 3);
    }
    g.inside = (uint32_t)lane < v.rays_per_task && g.px < v.width && g.py < v.height;
    g.pixel_id = (uint32_t)g.py * (uint32_t)v.width + (uint32_t)g.px;
    if (v.pixel_mask != nullptr && g.inside) g.inside = v.pixel_mask[g.pixel_id] != 0; // (parity tests: trace a chosen set of pixels; wave-uniform branch)
    return g;
}
// ray state of (task, lane): task-linear, rays_per_task entries per task (lanes beyond that own no ray and must not touch it)
EGR_DI StateRef state_of(const DeviceView &v, uint32_t task, int lane) { return StateRef{v.state, v.state_stride, task * v.rays_per_task + (uint32_t)lane}; }

} // namespace
