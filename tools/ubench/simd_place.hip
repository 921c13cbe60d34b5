// Where do the waves of a workgroup go? Reads HW_ID (s_getreg_b32 hwreg(HW_REG_HW_ID)) per wave: gfx9 layout wave_id[3:0] simd_id[5:4] pipe_id[7:6]
// cu_id[11:8] sh_id[12] se_id[15:13]. Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/ubench/simd_place.hip -o /tmp/simd_place && /tmp/simd_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep every workgroup resident for a while so that the launch fills the device like a persistent kernel does
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (id & 0xFFFFu) | (xcc << 16);
}
int main() {
    for (int waves : {4, 16}) {
        const int groups = 64;
        unsigned *d;
        hipMalloc(&d, groups * waves * 4);
        hipLaunchKernelGGL(k, dim3(groups), dim3(64 * waves), 0, 0, d, 20000);
        std::vector<unsigned> h(groups * waves);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("workgroups of %d waves: [xcc se.sh.cu : simd of wave 0, 1, ...]\n", waves);
        for (int g = 0; g < 12; g++) {
            unsigned a = h[g * waves];
            printf("  group %2d  xcc %u  se %u sh %u cu %2u : ", g, (a >> 16) & 15, (a >> 13) & 7, (a >> 12) & 1, (a >> 8) & 15);
            for (int w = 0; w < waves; w++) printf("%u", (h[g * waves + w] >> 4) & 3);
            bool same_cu = true;
            for (int w = 1; w < waves; w++) same_cu = same_cu && ((h[g * waves + w] >> 8) == (a >> 8));
            printf("  %s\n", same_cu ? "" : "(waves on several CUs)");
        }
        hipFree(d);
    }
    return 0;
}
