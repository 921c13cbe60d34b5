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
// task -> wave tile -> pixel. Tasks enumerate the 16x16 macro tiles owned by this rank (every world-th tile), 4 wave
// tiles each, in the order of v.task_macro: the task range is cut into 8 contiguous chunks (one per XCD queue), each
// chunk is a compact 2-D block of the image walked along a Z-curve, so the waves resident on one XCD at any time
// cover a compact patch (their rays meet the same BVH nodes / records -> per-XCD L2 hits, not fabric traffic).
EGR_DI TaskGeom task_geom(const DeviceView &v, uint32_t task, int lane) {
    uint32_t mtx = (uint32_t)(v.width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    uint32_t m = v.task_macro[task >> 2]; // this rank's macro tiles in cache-friendly order (see egr_build_task_order)
    uint32_t sub = task & 3u;
    int tx = (int)(m % mtx) * 2 + (int)(sub & 1u), ty = (int)(m / mtx) * 2 + (int)(sub >> 1);
    TaskGeom g;
    g.px = tx * EGR_TILE + (lane & 7);
    g.py = ty * EGR_TILE + (lane >> 3);
    g.inside = g.px < v.width && g.py < v.height;
    g.pixel_id = (uint32_t)g.py * (uint32_t)v.width + (uint32_t)g.px;
    return g;
}

} // namespace
