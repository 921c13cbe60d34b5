// Micro-benchmark: how should a lane fetch "its" random 128-B line?
//   mode 0: every lane issues 8 x 16-B loads of its own line               (what the BVH8 per-lane walk does)
//   mode 1: groups of 8 lanes cooperate: in sub-step i all 8 lanes read lane i's line, lane j takes slot j
//           (same bytes per lane, but each load instruction touches 8 lines instead of 64)
//   mode 2: every lane issues 1 x 16-B load of its own line                 (binary 16-B node walk)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(64) k(const uint4 *__restrict__ data, const uint32_t *__restrict__ idx, uint32_t *out, int iters, int mode, uint32_t nlines) {
    uint32_t lane = threadIdx.x, acc = 0;
    uint32_t cur = idx[(blockIdx.x * 64 + lane) % nlines];
    for (int it = 0; it < iters; it++) {
        if (mode == 0) {
            const uint4 *p = data + (size_t)cur * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) { uint4 v = p[k]; acc += v.x ^ v.w; }
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                uint32_t other = __shfl(cur, (lane & ~7u) | i);
                uint4 v = data[(size_t)other * 8 + (lane & 7u)];
                acc += v.x ^ v.w;
            }
        } else {
            uint4 v = data[(size_t)cur * 8]; acc += v.x ^ v.w;
        }
        cur = (cur * 1664525u + 1013904223u + acc) % nlines; // dependent next address, like a tree walk
    }
    out[blockIdx.x * 64 + lane] = acc;
}
int main() {
    const uint32_t nlines = 1u << 20; // 128 MB of 128-B lines (> L2, < MALL)
    uint4 *d; uint32_t *idx, *out;
    hipMalloc(&d, (size_t)nlines * 128); hipMemset(d, 1, (size_t)nlines * 128);
    std::vector<uint32_t> h(nlines); for (uint32_t i = 0; i < nlines; i++) h[i] = (uint32_t)rand() % nlines;
    hipMalloc(&idx, nlines * 4); hipMemcpy(idx, h.data(), nlines * 4, hipMemcpyHostToDevice);
    const int waves = 256 * 16, iters = 2000;
    hipMalloc(&out, waves * 64 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, idx, out, 50, mode, nlines);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, idx, out, iters, mode, nlines);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double lanes_it = (double)waves * 64 * iters;
        printf("mode %d: %.2f ms, %.2f G lane-iterations/s, %.2f TB/s of 128-B lines (mode 2: 16 B each)\n", mode, ms, lanes_it / ms / 1e6, lanes_it * (mode == 2 ? 16 : 128) / ms / 1e9);
    }
    return 0;
}
