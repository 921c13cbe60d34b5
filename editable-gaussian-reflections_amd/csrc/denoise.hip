// SPDX-License-Identifier: MIT
// Denoiser stand-in. The reference's `denoise()` runs the closed OptiX AI denoiser on `output_final` with
// `output_normal` as the guide layer and writes `output_denoised` (optix/denoiser_wrapper.h:42-59,75-105). That network cannot
// be reproduced, so this is the classic guided filter with the same inputs and output: the edge-avoiding a-trous wavelet
// transform (Dammertz, Sewtz, Hanika, Lensch 2010) - 5 passes of a 5x5 B3-spline kernel with holes 1, 2, 4, 8, 16, each tap
// weighted by colour and normal similarity, the colour tolerance halving from pass to pass. PARITY UNPINNED vs OptiX; the
// tests compare this kernel with a plain PyTorch fp32 implementation of the same filter (tests/test_denoise.py).
//
// One thread per pixel, 25 taps, everything in registers; the image is read through the L1/L2 (a 1080p float3 image is 25 MB:
// it lives in the 256 MB Infinity Cache between the passes), so each pass streams 2 x 25 MB.
#include <hip/hip_runtime.h>

#include "egr_internal.hpp"

namespace {

constexpr float SIGMA_COLOR = 0.6f;  // colour tolerance of the first pass (halved every pass)
constexpr float SIGMA_NORMAL = 0.3f; // normal tolerance (all passes)

__global__ void __launch_bounds__(256) k_atrous(int W, int H, int hole, float inv_sc2, float inv_sn2, const float *__restrict__ src,
                                                const float *__restrict__ normal, float *__restrict__ dst) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t p = ((size_t)y * W + x) * 3;
    const float cr = src[p], cg = src[p + 1], cb = src[p + 2];
    const float nx = normal[p], ny = normal[p + 1], nz = normal[p + 2];
    const float h[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, sw = 0.0f;
#pragma unroll
    for (int j = -2; j <= 2; j++) {
        const int yy = y + j * hole;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int i = -2; i <= 2; i++) {
            const int xx = x + i * hole;
            if (xx < 0 || xx >= W) continue;
            const size_t q = ((size_t)yy * W + xx) * 3;
            const float r = src[q], g = src[q + 1], b = src[q + 2];
            const float dr = r - cr, dg = g - cg, db = b - cb;
            const float dnx = normal[q] - nx, dny = normal[q + 1] - ny, dnz = normal[q + 2] - nz;
            const float wc = __expf(-(dr * dr + dg * dg + db * db) * inv_sc2);
            const float wn = __expf(-(dnx * dnx + dny * dny + dnz * dnz) * inv_sn2);
            const float w = h[i + 2] * h[j + 2] * wc * wn;
            sr += w * r, sg += w * g, sb += w * b, sw += w;
        }
    }
    const float inv = 1.0f / sw; // the centre tap always contributes 9/64
    dst[p] = sr * inv, dst[p + 1] = sg * inv, dst[p + 2] = sb * inv;
}

} // namespace

void egr_denoise_atrous(egr_context *c, hipStream_t s) {
    const int W = c->width, H = c->height;
    const size_t n = (size_t)W * H * 3;
    if (!c->denoise_tmp) egr_dev_alloc(c, c->denoise_tmp, 2 * n);
    float *tmp[2] = {c->denoise_tmp, c->denoise_tmp + n};
    const float *src = c->fb.output_final;
    const dim3 grid((W + 31) / 32, (H + 7) / 8), block(256);
    float sigma_c = SIGMA_COLOR;
    for (int pass = 0; pass < 5; pass++) {
        float *dst = pass == 4 ? c->fb.output_denoised : tmp[pass & 1];
        hipLaunchKernelGGL(k_atrous, grid, block, 0, s, W, H, 1 << pass, 1.0f / (sigma_c * sigma_c), 1.0f / (SIGMA_NORMAL * SIGMA_NORMAL), src,
                           (const float *)c->fb.output_normal, dst);
        src = dst;
        sigma_c *= 0.5f;
    }
}
