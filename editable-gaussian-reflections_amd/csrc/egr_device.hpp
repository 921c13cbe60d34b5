// Device-side arithmetic of the hot path (gfx950). Each function cites the reference expression it
// implements (paths relative to /root/reference/editable_gauss_refl/cuda/csrc/). Written independently of
// oracle/: the oracle restates the reference for testing; this header is the product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EGR_DI __device__ __forceinline__

struct f3 {
    float x, y, z;
};
#define V3 f3
#define MK3 mk3
#include "egr_vec.inc"
#undef V3
#undef MK3
EGR_DI bool finite3(f3 a) { return isfinite(a.x + a.y + a.z); }

// ---- RNG: utils/random.h:34-62 --------------------------------------------------------------------------
EGR_DI uint32_t tea4(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
EGR_DI float rnd(uint32_t &prev) {
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}

// ---- activations: utils/activations.cu ------------------------------------------------------------------
EGR_DI float sigmoid_act(float x) { return 1.0f / (1.0f + expf(-x)); }
EGR_DI float relu_act(float x) { return fmaxf(0.0f, x); }
EGR_DI float clip01_act(float x) { return fminf(fmaxf(0.0f, x), 1.0f); }
EGR_DI float sign1(float v) { return copysignf(1.0f, v); } // utils/misc.cu:59-63, sign(0) = +1

// ---- kernel.cu:3-16 -------------------------------------------------------------------------------------
EGR_DI float compute_scaling_factor(float opacity, float alpha_threshold, float exp_power) {
    float k = 2.0f * exp_power;
    return opacity <= alpha_threshold ? 0.0f : powf(k * logf(opacity / alpha_threshold), 1.0f / k);
}
EGR_DI float pow_exp(float d, float p) { return p == 3.0f ? d * d * d : powf(d, p); }
EGR_DI float pow_exp_m1(float d, float p) { return p == 3.0f ? d * d : powf(d, p - 1.0f); }
EGR_DI float eval_gaussian_sq(float sq, float exp_power) { return expf(-pow_exp(sq, exp_power) / (2.0f * exp_power)); }

// ---- misc ---------------------------------------------------------------------------------------------
EGR_DI uint32_t f2u(float f) { return __float_as_uint(f); }
EGR_DI float u2f(uint32_t u) { return __uint_as_float(u); }

#pragma clang fp contract(off)
// IEEE square root and division of the hot per-candidate / per-hit / per-ray arithmetic WITHOUT the range scaling the compiler's expansions carry (v_div_scale / v_div_fixup,
// the 2^32 pre-scaling of a denormal radicand): the same correction steps on the same hardware approximations - the sequence the compiler emits
// for `sqrtf(x)` and `a / b` with the scaling taken out, so the results are the correctly rounded ones whenever no intermediate leaves the normal
// range (|W d| of a usable gaussian is within 1e-15 ... 1e15). 17 of the 75 instructions of a rejected candidate; quotients that share a
// denominator share its refined reciprocal.
#if defined(__HIP_DEVICE_COMPILE__)
EGR_DI float egr_sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x); // within 1 ulp
    const float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);
    const float t = __builtin_fmaf(-sd, s, x) <= 0.0f ? sd : s;
    return __builtin_fmaf(-su, s, x) > 0.0f ? su : t;
}
EGR_DI float egr_rcp_refined(float b) { // the reciprocal both quotients below start from
    const float r0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
}
EGR_DI float egr_div_rn(float a, float b, float r) { // a / b, r = egr_rcp_refined(b)
    const float q0 = a * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), r, q1);
}
#else // (host pass of the translation unit)
EGR_DI float egr_sqrt_rn(float x) { return sqrtf(x); }
EGR_DI float egr_rcp_refined(float) { return 0.0f; }
EGR_DI float egr_div_rn(float a, float b, float) { return a / b; }
#endif
#pragma clang fp contract(fast)

// Ray segment [tmin,tmax] vs axis-aligned box with precomputed 1/d; NaN-free for finite rays because
// fminf/fmaxf drop a NaN operand (0*inf when the origin lies on a slab plane of a zero direction component).
EGR_DI bool slab_hit(f3 lo, f3 hi, f3 o, f3 inv, float tmin, float tmax) {
    float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return t0 <= t1;
}
