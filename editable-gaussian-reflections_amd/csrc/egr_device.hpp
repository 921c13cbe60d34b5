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
EGR_DI f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
EGR_DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
EGR_DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
EGR_DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
EGR_DI f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
EGR_DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
EGR_DI f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
EGR_DI f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
EGR_DI f3 div_s(f3 a, float s) { // sutil float3/float = a * (1/s), utils/vec_math.h:330-333
    float inv = 1.0f / s;
    return a * inv;
}
EGR_DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
EGR_DI f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
EGR_DI float length(f3 a) { return sqrtf(dot(a, a)); }
EGR_DI f3 normalize(f3 v) { // utils/vec_math.h:376-379
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
EGR_DI f3 reflect(f3 i, f3 n) { return i - 2.0f * n * dot(n, i); } // utils/vec_math.h:385
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

// ---- GGX: utils/ggx_brdf.h (only the functions the reference calls) -----------------------------------
// BRDF_EPS and M_PI are double literals upstream (:6), so the affected sums are done in double here too.
// These are only called from epilogue.hip, which is compiled with -ffp-contract=off (see the note there).
EGR_DI float G1(f3 N, f3 W, float alpha) { // :32-37
    float NdotW = fmaxf(dot(N, W), 0.0f);
    float k = (alpha * alpha) / 2.0f;
    float partial = NdotW * (1.0f - k) + k;
    return (float)((double)NdotW / ((double)partial + 1e-8));
}
EGR_DI f3 cook_torrance_weight(f3 N, f3 V, f3 L, float roughness, f3 f0) { // :134-150
    if (f0.x == 0.0f && f0.y == 0.0f && f0.z == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    f3 H = normalize(V + L);
    float NdotH = fmaxf(dot(N, H), 0.0f);
    float VdotH = fmaxf(dot(V, H), 0.0f);
    float NdotV = fmaxf(dot(N, V), 0.0f);
    float alpha = roughness * roughness;
    float G = G1(N, V, alpha) * G1(N, L, alpha);
    float p = powf(1.0f - VdotH, 5.0f); // fresnel_schlick :83
    f3 F = mk3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
    float denom = (float)((double)(NdotH * NdotV) + 1e-8);
    return div_s((F * G) * VdotH, denom);
}
EGR_DI f3 sample_cook_torrance(f3 N, f3 V, float roughness, float u1, float u2) { // :152-168
    float alpha = roughness * roughness;
    float phi = (float)(2.0 * 3.14159265358979323846 * (double)u1);
    float cosTheta = sqrtf((1.0f - u2) / (1.0f + (alpha * alpha - 1.0f) * u2));
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    f3 Hl = mk3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    f3 up = (N.z < 0.999f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f); // :163 (sic: N.z, not |N.z|)
    f3 T = normalize(cross(up, N));
    f3 B = cross(N, T);
    f3 H = Hl.x * T + Hl.y * B + Hl.z * N;
    return reflect(-V, H);
}

// ---- misc ---------------------------------------------------------------------------------------------
EGR_DI uint32_t f2u(float f) { return __float_as_uint(f); }
EGR_DI float u2f(uint32_t u) { return __uint_as_float(u); }

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
