// Micro-benchmark: rate of non-returning fp32 atomic adds issued as 64-B records (16 lanes x 4 B to one gradient row), the form in which
// the backward chain sends its gradients (wide_add_wave, trace.hip). Prints G lane-atomics / s for row sets of different sizes
// (L2-resident ... HBM-resident). Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void __launch_bounds__(64) k_records(float *rows, uint32_t row_mask, uint32_t iters, uint32_t lanes_per_record) {
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t s = wave * 2654435761u + 12345u;
    for (uint32_t i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t rec = lane / lanes_per_record, comp = lane % lanes_per_record;
        const uint32_t row = ((s >> 4) + rec * 0x9E3779B1u) * 2246822519u >> 8 & row_mask; // one pseudo-random row per record
        atomicAdd(rows + (size_t)row * 32 + comp, 1.0f);
    }
}

int main() {
    const uint32_t waves = 256 * 12, iters = 4000;
    float *rows;
    const size_t max_rows = 1u << 21; // 2M rows x 128 B = 256 MB
    (void)hipMalloc(&rows, max_rows * 128 + 4096);
    (void)hipMemset(rows, 0, max_rows * 128 + 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (uint32_t lpr : {16u, 64u, 1u}) {
        for (uint32_t bits : {10u, 14u, 17u, 20u, 21u}) {
            const uint32_t mask = (1u << bits) - 1u;
            hipLaunchKernelGGL(k_records, dim3(waves), dim3(64), 0, 0, rows, mask, 200u, lpr); // warm-up
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_records, dim3(waves), dim3(64), 0, 0, rows, mask, iters, lpr);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double lane_atomics = (double)waves * iters * 64.0;
            printf("lanes per record %2u, rows %8u (%7.1f MB): %8.3f ms  %7.1f G lane-atomics/s  %7.1f G records/s\n", lpr, mask + 1, (mask + 1) * 128.0 / 1e6, ms,
                   lane_atomics / ms / 1e6, lane_atomics / lpr / ms / 1e6);
        }
    }
    return 0;
}
