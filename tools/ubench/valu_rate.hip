// How many cycles does one wave64 fp32 VALU instruction occupy a SIMD on this GPU? (tools/ubench; hipcc --offload-arch=gfx950 -O3)
// W waves per SIMD each run a long stream of independent v_fma_f32 (8 accumulators); time -> wave-instructions per second per SIMD.
// Also v_cvt_f32_u32 (SDWA word select), v_min_f32 and a mixed slab-test-like body.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 1) {
#define F(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 2) {
#define F(x) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 3) {
#define F(x, y) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&x) : "v"(*(double *)&y), "v"(*(double *)&y));
            double d0 = x0, d1 = x1, d2 = x2, d3 = x3, dy = a;
            F(d0, dy) F(d1, dy) F(d2, dy) F(d3, dy) F(d0, dy) F(d1, dy) F(d2, dy) F(d3, dy) F(d0, dy) F(d1, dy) F(d2, dy) F(d3, dy) F(d0, dy) F(d1, dy) F(d2, dy) F(d3, dy)
            x0 += (float)d0, x1 += (float)d1, x2 += (float)d2, x3 += (float)d3;
#undef F
        } else if (MODE == 4) {
#define F(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 5) {
#define F(x) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(x));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7) F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        }
#define OP2(M, TXT)                                                                                       \
        else if (MODE == M) {                                                                             \
            _Pragma("unroll") for (int r = 0; r < 2; r++) {                                                \
                asm volatile(TXT " %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile(TXT " %0, %0, %1" : "+v"(x1) : "v"(a)); \
                asm volatile(TXT " %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile(TXT " %0, %0, %1" : "+v"(x3) : "v"(a)); \
                asm volatile(TXT " %0, %0, %1" : "+v"(x4) : "v"(a)); asm volatile(TXT " %0, %0, %1" : "+v"(x5) : "v"(a)); \
                asm volatile(TXT " %0, %0, %1" : "+v"(x6) : "v"(a)); asm volatile(TXT " %0, %0, %1" : "+v"(x7) : "v"(a)); \
            }                                                                                             \
        }
        OP2(10, "v_add_f32") OP2(11, "v_mul_f32") OP2(12, "v_max_f32") OP2(13, "v_and_b32") OP2(14, "v_lshrrev_b32") OP2(15, "v_add_u32") OP2(16, "v_sub_f32")
        OP2(17, "v_or_b32") OP2(18, "v_mul_u32_u24") OP2(19, "v_min_u32") OP2(20, "v_mul_lo_u32") OP2(21, "v_fmac_f32") OP2(22, "v_min_i32") OP2(23, "v_xor_b32")
#define OP3(M, TXT)                                                                                       \
        else if (MODE == M) {                                                                             \
            _Pragma("unroll") for (int r = 0; r < 2; r++) {                                                \
                asm volatile(TXT " %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b)); asm volatile(TXT " %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b)); \
                asm volatile(TXT " %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b)); asm volatile(TXT " %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b)); \
                asm volatile(TXT " %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b)); asm volatile(TXT " %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b)); \
                asm volatile(TXT " %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b)); asm volatile(TXT " %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b)); \
            }                                                                                             \
        }
        OP3(30, "v_med3_f32") OP3(31, "v_min3_f32") OP3(32, "v_lshl_add_u32") OP3(33, "v_and_or_b32") OP3(34, "v_mad_u32_u24") OP3(35, "v_bfe_u32") OP3(36, "v_add3_u32") OP3(37, "v_perm_b32")
        else if (MODE == 40) { // compare into vcc + select
#define F(x) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 41) { // compare only (two per accumulator pair)
#define F(x) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 42) { // v_cvt_f32_ubyte0 / plain cvt
#define F(x) asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_u32 %0, %0" : "+v"(x));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 43) { // transcendental pair
#define F(x) asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %0, %0" : "+v"(x));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        } else if (MODE == 44) { // DPP move + readlane-free mov
#define F(x) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32 %0, %0" : "+v"(x));
            F(x0) F(x1) F(x2) F(x3) F(x4) F(x5) F(x6) F(x7)
#undef F
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char *name, float *out, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4, blocks = simds * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * iters * 16;
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f G wave-instr/s per chip, %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms, insts / ms / 1e6,
           2.4e9 * simds / (insts / (ms * 1e-3)));
}
int main() {
    float *out;
    hipMalloc(&out, 64 << 20);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", out, w), run<1>("v_min_f32", out, w), run<2>("v_cvt_f32_u32_sdwa", out, w), run<3>("v_pk_fma_f32", out, w), run<4>("v_max3_f32", out, w), run<5>("v_mbcnt_lo", out, w);
    }
    const int w = 4;
    run<10>("v_add_f32", out, w), run<11>("v_mul_f32", out, w), run<16>("v_sub_f32", out, w), run<21>("v_fmac_f32", out, w), run<12>("v_max_f32", out, w), run<13>("v_and_b32", out, w), run<17>("v_or_b32", out, w), run<23>("v_xor_b32", out, w);
    run<14>("v_lshrrev_b32", out, w), run<15>("v_add_u32", out, w), run<18>("v_mul_u32_u24", out, w), run<19>("v_min_u32", out, w), run<22>("v_min_i32", out, w), run<20>("v_mul_lo_u32", out, w);
    run<30>("v_med3_f32", out, w), run<31>("v_min3_f32", out, w), run<32>("v_lshl_add_u32", out, w), run<33>("v_and_or_b32", out, w), run<34>("v_mad_u32_u24", out, w), run<35>("v_bfe_u32", out, w), run<36>("v_add3_u32", out, w), run<37>("v_perm_b32", out, w);
    run<40>("v_cmp_le_f32+v_cndmask (x8: halve)", out, w), run<41>("v_cmp x2 (x8: halve)", out, w), run<42>("cvt_ubyte1+cvt_u32 (halve)", out, w), run<43>("v_rcp+v_sqrt (halve)", out, w), run<44>("mov_dpp+mov (halve)", out, w);
    return 0;
}
