// Per-ray step epilogue = forward_pass.cu:142-155 (tail renormalisation, R4) + shaders.cu:111-147 (throughput,
// GGX bounce sampling, next ray, R5). One thread per ray, once per bounce step.
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off (see build.py). sample_cook_torrance evaluates
// sin = sqrt(1 - cos*cos) with cos ~ 1 for near-mirror roughness; fusing that product into an fma changes the
// sampled direction by ~1e-4, which then decides which Gaussians the bounce ray meets. Evaluating the handful of
// per-ray operations unfused (plain IEEE mul/add, what the source text says) keeps the bounce rays
// bit-comparable with the CPU oracle. It costs ~100 extra instructions per ray-step; the per-candidate hot loops
// in trace.hip keep fma contraction.
#include <algorithm>

#include "egr_state.hpp"

namespace {

__global__ void __launch_bounds__(EGR_WAVE) k_step_epilogue(DeviceView v, int step, int grads) {
    const int lane = threadIdx.x;
    const int num_bounces = min(*v.cfg.num_bounces, EGR_MAX_BOUNCES);
    if (step > num_bounces) return;
    for (uint32_t task = v.task_begin + blockIdx.x; task < v.task_begin + v.task_count; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        if (!tg.inside) continue;
        StateRef S{v.state, v.state_stride, task * EGR_WAVE + (uint32_t)lane};
        if (step > 0 && S.ld(F_ALIVE) == 0.0f) continue;
        const f3 ro = S.ld3(F_RAY_O), rd = S.ld3(F_RAY_D);
        uint32_t seed = f2u(S.ld(F_SEED));
        const f3 c_rgb = S.ld3(SF(step, S_RGB)), c_n = S.ld3(SF(step, S_NORMAL)), c_f0 = S.ld3(SF(step, S_F0));
        const float c_depth = S.ld(SF(step, S_DEPTH)), c_rough = S.ld(SF(step, S_ROUGH)), T = S.ld(SF(step, S_T)), full_T = S.ld(SF(step, S_TTOT));

        // ---- R4: forward_pass.cu:142-155 ----
        const float rem = T - full_T;
        const float normalization = fmaxf(1.0f - T, *v.cfg.eps_forward_normalization);
        const f3 r_rgb = div_s(c_rgb, normalization), r_n = div_s(c_n, normalization), r_f0 = div_s(c_f0, normalization);
        const float r_depth = c_depth / normalization, r_rough = c_rough / normalization;
        f3 o_rgb = c_rgb + rem * r_rgb;
        const f3 o_n = c_n + rem * r_n, o_f0 = c_f0 + rem * r_f0;
        const float o_depth = c_depth + rem * r_depth, o_rough = c_rough + rem * r_rough;

        // ---- R5: shaders.cu:111-147 ----
        f3 thr_prev = mk3(1, 1, 1);
        if (step > 0) {
            thr_prev = S.ld3(SF(step - 1, S_THR));
            o_rgb = o_rgb * thr_prev; // :112-114
        }
        const f3 eff_n = normalize(o_n);
        const float eff_rough = fmaxf(o_rough, *v.cfg.eps_min_roughness);
        const bool cont = !(length(o_n) < *v.cfg.reflection_invalid_normal_threshold); // :123
        f3 next_o = mk3(0, 0, 0), next_d = mk3(0, 0, 0), thr = mk3(1, 1, 1);
        if (cont) {
            const f3 pos = ro + o_depth * rd;
            const float u1 = rnd(seed); // make_float2(rnd(seed), rnd(seed)): evaluated left to right
            const float u2 = rnd(seed);
            next_d = sample_cook_torrance(eff_n, -rd, eff_rough, u1, u2);
            next_o = pos + *v.cfg.eps_ray_surface_offset * next_d;
            thr = thr_prev * cook_torrance_weight(eff_n, -rd, next_d, eff_rough, o_f0); // :134-140
        }
        S.st3(SF(step, S_RGB), o_rgb), S.st(SF(step, S_DEPTH), o_depth), S.st3(SF(step, S_NORMAL), o_n);
        S.st3(SF(step, S_F0), o_f0), S.st(SF(step, S_ROUGH), o_rough);
        if (grads) {
            S.st3(SF(step, S_REM_RGB), r_rgb), S.st(SF(step, S_REM_DEPTH), r_depth), S.st3(SF(step, S_REM_NORMAL), r_n);
            S.st3(SF(step, S_REM_F0), r_f0), S.st(SF(step, S_REM_ROUGH), r_rough);
        }
        S.st3(SF(step, S_THR), thr), S.st3(SF(step, S_NEXT_O), next_o), S.st3(SF(step, S_NEXT_D), next_d);
        S.st(F_STEPS, u2f((uint32_t)step + 1u));
        S.st(F_ALIVE, (cont && step < num_bounces) ? 1.0f : 0.0f);
        S.st3(F_RAY_O, next_o), S.st3(F_RAY_D, next_d);
        S.st(F_SEED, u2f(seed));
        v.meta.random_seeds[tg.pixel_id] = (int32_t)seed; // shaders.cu:172
    }
}

} // namespace

void egr_launch_step_epilogue(const DeviceView &v, int step, bool grads, hipStream_t s) {
    hipLaunchKernelGGL(k_step_epilogue, dim3(std::max(1u, std::min(v.task_count, 65535u))), dim3(EGR_WAVE), 0, s, v, step, grads ? 1 : 0);
}
