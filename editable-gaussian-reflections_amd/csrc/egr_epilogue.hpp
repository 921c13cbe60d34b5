// Per-ray step epilogue = forward_pass.cu:142-155 (tail renormalisation, R4) + shaders.cu:111-147 (throughput,
// GGX bounce sampling, next ray, R5). One lane per ray, once per bounce step.
//
// EVERYTHING IN THIS HEADER IS COMPILED WITH FMA CONTRACTION OFF (the pragma below). sample_cook_torrance evaluates sin = sqrt(1 - cos*cos) with cos ~ 1 for near-mirror roughness; fusing
// that product into an fma changes the sampled direction by ~1e-4, which then decides which Gaussians the bounce ray
// meets. Evaluating the handful of per-ray operations unfused (plain IEEE mul/add, what the source text says) keeps the
// bounce rays bit-comparable with the CPU oracle. The vector type is a distinct one (f3u) so that its operators are the
// contraction-free instantiation of egr_vec.inc inside the fused per-tile chain of trace.hip. It costs ~100 extra instructions per ray-step; the per-candidate
// hot loops keep contraction.
#pragma once
#include "egr_state.hpp"

#pragma clang fp contract(off)
struct f3u {
    float x, y, z;
};
#define V3 f3u
#define MK3 mk3u
#include "egr_vec.inc"
#undef V3
#undef MK3

namespace {
EGR_DI f3u unfused(f3 a) { return f3u{a.x, a.y, a.z}; }
EGR_DI f3 fused(f3u a) { return f3{a.x, a.y, a.z}; }

// the lane's ray of `task` after forward step `step` (the caller checked tg.inside)
EGR_DI void step_epilogue_lane(const DeviceView &v, int step, bool grads, int num_bounces, const TaskGeom &tg, const StateRef &S) {
    if (step > 0 && S.ld(F_ALIVE) == 0.0f) return;
    const f3u ro = unfused(S.ld3(F_RAY_O)), rd = unfused(S.ld3(F_RAY_D));
    uint32_t seed = f2u(S.ld(F_SEED));
    const f3u c_rgb = unfused(S.ld3(SF(step, S_RGB))), c_n = unfused(S.ld3(SF(step, S_NORMAL))), c_f0 = unfused(S.ld3(SF(step, S_F0)));
    const float c_depth = S.ld(SF(step, S_DEPTH)), c_rough = S.ld(SF(step, S_ROUGH)), T = S.ld(SF(step, S_T)), full_T = S.ld(SF(step, S_TTOT));

    // ---- R4: forward_pass.cu:142-155 ----
    const float rem = T - full_T;
    const float normalization = fmaxf(1.0f - T, *v.cfg.eps_forward_normalization);
    // (sutil's float3 / float is a * (1 / s), utils/vec_math.h:330-333: ONE reciprocal for the three vectors, two true quotients for the scalars - all
    // three from one refined reciprocal of the denominator, the results of `/`: egr_device.hpp)
    const float rnorm = egr_rcp_refined(normalization), inv_norm = egr_div_rn(1.0f, normalization, rnorm);
    const f3u r_rgb = c_rgb * inv_norm, r_n = c_n * inv_norm, r_f0 = c_f0 * inv_norm;
    const float r_depth = egr_div_rn(c_depth, normalization, rnorm), r_rough = egr_div_rn(c_rough, normalization, rnorm);
    f3u o_rgb = c_rgb + rem * r_rgb;
    const f3u o_n = c_n + rem * r_n, o_f0 = c_f0 + rem * r_f0;
    const float o_depth = c_depth + rem * r_depth, o_rough = c_rough + rem * r_rough;

    // ---- R5: shaders.cu:111-147 ----
    f3u thr_prev = mk3u(1, 1, 1);
    if (step > 0) {
        thr_prev = unfused(S.ld3(SF(step - 1, S_THR)));
        o_rgb = o_rgb * thr_prev; // :112-114
    }
    const f3u eff_n = normalize(o_n);
    const float eff_rough = fmaxf(o_rough, *v.cfg.eps_min_roughness);
    const bool cont = !(length(o_n) < *v.cfg.reflection_invalid_normal_threshold); // :123
    f3u next_o = mk3u(0, 0, 0), next_d = mk3u(0, 0, 0), thr = mk3u(1, 1, 1);
    if (cont) {
        const f3u pos = ro + o_depth * rd;
        const float u1 = rnd(seed); // make_float2(rnd(seed), rnd(seed)): evaluated left to right
        const float u2 = rnd(seed);
        next_d = sample_cook_torrance(eff_n, -rd, eff_rough, u1, u2);
        next_o = pos + *v.cfg.eps_ray_surface_offset * next_d;
        thr = thr_prev * cook_torrance_weight(eff_n, -rd, next_d, eff_rough, o_f0); // :134-140
    }
    S.st3(SF(step, S_RGB), fused(o_rgb)), S.st(SF(step, S_DEPTH), o_depth), S.st3(SF(step, S_NORMAL), fused(o_n));
    S.st3(SF(step, S_F0), fused(o_f0)), S.st(SF(step, S_ROUGH), o_rough);
    if (grads) {
        S.st3(SF(step, S_REM_RGB), fused(r_rgb)), S.st(SF(step, S_REM_DEPTH), r_depth), S.st3(SF(step, S_REM_NORMAL), fused(r_n));
        S.st3(SF(step, S_REM_F0), fused(r_f0)), S.st(SF(step, S_REM_ROUGH), r_rough);
    }
    S.st3(SF(step, S_THR), fused(thr)), S.st3(SF(step, S_NEXT_O), fused(next_o)), S.st3(SF(step, S_NEXT_D), fused(next_d));
    S.st(F_STEPS, u2f((uint32_t)step + 1u));
    S.st(F_ALIVE, (cont && step < num_bounces) ? 1.0f : 0.0f);
    S.st3(F_RAY_O, fused(next_o)), S.st3(F_RAY_D, fused(next_d));
    S.st(F_SEED, u2f(seed));
    v.meta.random_seeds[tg.pixel_id] = (int32_t)seed; // shaders.cu:172
}
} // namespace
#ifdef EGR_CONTRACT_AFTER_EPILOGUE // trace.hip: back to the translation unit's default for what follows
#pragma clang fp contract(fast)
#endif
