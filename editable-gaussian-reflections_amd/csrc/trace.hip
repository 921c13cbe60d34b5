// The hot path: __raygen__rg + __intersection__gaussian + forward_pass + backward_pass of the reference
// (shaders.cu:9-173, forward_pass.cu:3-155, backward_pass.cu:3-222) as hand-written HIP kernels for gfx950.
//
// Structure (MI355X-first; nothing here is a translation of the OptiX pipeline; DESIGN.md 4 has the measurements):
//   * one wave64 = one 8x8 pixel tile, one lane = one ray; PERSISTENT waves pull tiles from 8 XCD-affine atomic queues
//     (ragged per-ray work: 1..1000s of candidates); waves never wait on each other (no workgroup barrier after the first instruction);
//   * a tile runs through ALL its steps in one go (k_forward_chain: trace, step epilogue, trace, ...; k_backward_chain), so
//     a launch pays the tail of its heaviest tile once per chain, not once per step. The per-step source lives in
//     forward_task.inc / backward_task.inc / egr_epilogue.hpp. Per-ray state lives in a task-linear SoA buffer (fully coalesced);
//   * the 8-wide BVH (128-B line-sized nodes, 16-bit child boxes) is walked by PRIMARY tiles as one FRUSTUM (all 64 rays leave the camera
//     origin: one interval test per child slot for the whole tile, eight nodes per iteration; the leaves found are evaluated one after
//     the other, records through the scalar cache, every lane for its own ray) and by BOUNCE tiles as PAIRS (one wave-wide LIFO of
//     (ray, node) pairs and one buffer of (ray, leaf) pairs in LDS: lane m of group g tests child slot m of the g-th popped node, ballots
//     compact the hits, and 64 leaf pairs at a time are evaluated with one lane per pair) - pair_walk below, frustum walk in forward_task.inc;
//   * the forward chain exists as single-wave workgroups and as TEAMS of waves with a shared LDS in which waves without tiles walk pairs
//     their team mates offer (egr_set_team_help, on by default: several waves on one heavy tile - the tail of every launch, most of an under-filled rank's);
//   * the reference's global per-pixel linked list (one same-address atomic per candidate, 36 B entries, pointer chasing)
//     is replaced by a per-resident-wave candidate scratch, one contiguous run per lane ([lane][k]) with bump-allocated
//     extension blocks for the rare long list; it is reused tile after tile;
//   * depth ordering = strict-successor selection in batches of 8 (sorted insertion in registers, stable on ties);
//     semantically the reference's 16-at-a-time k-buffer (forward_pass.cu:55-137) INCLUDING its batch-boundary tie drop (Q3: the tie
//     continuation is switched off at every second batch boundary, forward_task.inc);
//   * composited hits needed by the backward pass go to an arena in 8-row blocks (one atomic per 8 rows per WAVE instead
//     of one per hit per lane), chained newest->oldest, which is the order backward walks;
//   * backward recomputes the local hit point from the snapshot transform instead of storing it. Primary tiles sum their
//     contributions per gaussian in an LDS hash table (DPP neighbour pre-reduction first); whatever leaves the wave - a
//     bounce hit, a flushed table slot - is added to its gaussian's position-ordered gradient row by SIXTEEN LANES PER
//     RECORD (one 64-B non-returning atomic request instead of 15 scattered ones; wide_add_wave); k_grad_gather scatters
//     the rows to the reference's gradient tensors.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#define EGR_CONTRACT_AFTER_EPILOGUE
#include "egr_epilogue.hpp" // (includes egr_state.hpp) the step epilogue, for the fused per-tile chain

#ifndef EGR_GPOP
#define EGR_GPOP 5 // pair walk: a walk batch pops up to 8 x EGR_GPOP (ray, node) pairs, EGR_GPOP per lane group (5 since the loop is instantiated per decode: 62 -> 41 spilled VGPRs; before that 4 was best)
#endif
#define EGR_WALK_MAX_LEAVES (8 * EGR_GPOP * EGR_WIDTH) // leaf pairs one walk batch can add
#define EGR_LBUF (64 + 2 * EGR_WALK_MAX_LEAVES) // (an evaluation batch and a walk batch are issued together: room for two walk batches' leaves)
#ifndef EGR_FPOP
#define EGR_FPOP 2 // frustum walk: 8 x EGR_FPOP nodes per iteration (two interleaved same-box pairs: forward chain 3.80-3.87 against 3.92-3.95 ms dense-init; 3 and 4 no further - and the ray table holds the origins of at most 64 + 8 x 2 x 8 buffered leaves: a static_assert says so)
#endif
#ifndef EGR_PSTK
#define EGR_PSTK 512 // pair-stack entries kept in LDS
#endif

namespace {

// XCD-affine persistent scheduling: the task range is cut into 8 contiguous chunks (contiguous image bands), one
// queue head per chunk. A wave first drains the chunk of "its" XCD (workgroup b is observed to run on XCD b % 8;
// used for speed only, any placement is correct), then steals from the others. Each XCD's private 4 MiB L2 then
// holds one band's BVH nodes / Gaussian records instead of the whole frame's, and the single shared head no
// longer serialises 6144 pullers (microarch guide, "dequeue": shard the head per XCD above 64 pullers).
EGR_DI uint32_t wave_next_task(uint32_t *heads, uint32_t num_tasks, uint32_t &cur_q, int lane) {
    const uint32_t chunk = ((num_tasks + 7u) / 8u + 3u) & ~3u;
    for (uint32_t tries = 0; tries < 8u; tries++) {
        const uint32_t q = (cur_q + tries) & 7u;
        const uint32_t beg = q * chunk, end = min(beg + chunk, num_tasks);
        if (beg >= end) continue;
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(heads + q * EGR_QUEUE_STRIDE, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (beg + t < end) {
            cur_q = q;
            return beg + t;
        }
    }
    return 0xFFFFFFFFu;
}
EGR_DI uint32_t wave_sum_u32(uint32_t x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off);
    return x;
}
EGR_DI uint32_t wave_max_u32(uint32_t x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, off));
    return x;
}
EGR_DI void add64(uint32_t *ctrl, int word, uint32_t x) {
    if (x) atomicAdd(reinterpret_cast<unsigned long long *>(ctrl + word), (unsigned long long)x);
}

// T4: primary ray (core/camera.h:17-36). Draw order: jitter.x then jitter.y.
// `view_size` = tanf(vertical_fov / 2) (core/camera.h:21), a launch constant that the compiler cannot hoist out of a task loop (a ~ 150-instruction range
// reduction of a value read through a pointer). The BACKWARD chain evaluates it once in front of its task loop (-1.2 %: 2.37-2.39 against 2.40-2.41 ms, two
// interleaved same-box pairs); the forward chain keeps it per tile: held across its task loop the value costs two more spilled VGPRs and 2 % (2.94-2.99 against
// 2.89-2.90 ms), and read per tile from a word the prologue wrote it measures the same as the tanf (2.85-2.87 against 2.84-2.86 ms) - as long as that word sits on a
// cache line of its own: on the line of the arena's bump counter the one load per tile took the chain from 2.9 to 4.5 ms.
EGR_DI f3 primary_direction(const DeviceView &v, int ix, int iy, bool jitter, uint32_t &seed, const float view_size) {
    float aspect_ratio = (float)v.width / (float)v.height;
    float fx = (float)ix, fy = (float)iy;
    if (jitter) {
        float jx = rnd(seed) - 0.5f;
        float jy = rnd(seed) - 0.5f;
        fx += jx;
        fy += jy;
    }
    float y = view_size * (1.0f - 2.0f * (fy + 0.5f) / (float)v.height);
    float x = aspect_ratio * view_size * (2.0f * (fx + 0.5f) / (float)v.width - 1.0f);
    const float *w = v.cam.rotation_w2c; // rows of w2c (= columns of c2w)
    f3 w0 = mk3(w[0], w[1], w[2]), w1 = mk3(w[3], w[4], w[5]), w2 = mk3(w[6], w[7], w[8]);
    return normalize(w0 * x + w1 * y - w2);
}

// The geometry of one (ray, gaussian) pair - object-space ray, closest-approach point, the two rejections that only need those
// (shaders.cu:19-20, 36, 41-51) - in ONE function shared by the forward's candidate test and the backward's recomputation (bit-identical
// t and u on both sides), with fused multiply-adds (nvcc's default for the reference too; the CPU oracle evaluates unfused, -ffp-contract=off).
// Which products of `a b + c d + e f` are fused is the compiler's choice PER CALL SITE (it differs between a uniform and a per-lane origin, and between
// two instantiations of one template), while the forward's primary tiles, its pair walk - owner's and helpers' copies - and the backward's
// recomputation must agree to the bit: the fusion is spelled out (the pattern the compiler applied to these expressions in rounds 1-5:
// fma(e, f, fma(a, b, c d))) and the functions themselves are compiled without contraction.
#pragma clang fp contract(off)
EGR_DI float egr_dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ax, bx, ay * by));
}
EGR_DI float egr_madd(float a, float b, float c) { // a b + c
    return __builtin_fmaf(a, b, c);
}
EGR_DI f3 egr_div3_rn(const f3 &a, const f3 &b, const f3 &r) { return mk3(egr_div_rn(a.x, b.x, r.x), egr_div_rn(a.y, b.y, r.y), egr_div_rn(a.z, b.z, r.z)); } // a / b per component, r = refined reciprocals of b
EGR_DI float egr_gaussian_sq(float sq, float exp_power) { // eval_gaussian_sq (kernel.cu:8-12) with the division by the launch constant 2p spelled out: its reciprocal is loop-invariant
    const float two_p = 2.0f * exp_power;
    return expf(egr_div_rn(-pow_exp(sq, exp_power), two_p, egr_rcp_refined(two_p)));
}
// (the object-space ORIGIN of a ray is the same for every ray that leaves one point: primary tiles compute it once per leaf, forward_task.inc)
EGR_DI f3 object_origin(const float4 &w0, const float4 &w1, const float4 &w2, const f3 &o) {
    f3 lo;
    lo.x = egr_dot3(w0.x, w0.y, w0.z, o.x, o.y, o.z) + w0.w, lo.y = egr_dot3(w1.x, w1.y, w1.z, o.x, o.y, o.z) + w1.w, lo.z = egr_dot3(w2.x, w2.y, w2.z, o.x, o.y, o.z) + w2.w;
    return lo;
}
EGR_DI void candidate_geometry_from(const float4 &w0, const float4 &w1, const float4 &w2, const f3 &lo, const f3 &d, f3 &ld, f3 &dhat, float &t, f3 &u, bool &behind,
                                    bool &outside) {
    ld.x = egr_dot3(w0.x, w0.y, w0.z, d.x, d.y, d.z), ld.y = egr_dot3(w1.x, w1.y, w1.z, d.x, d.y, d.z), ld.z = egr_dot3(w2.x, w2.y, w2.z, d.x, d.y, d.z);
    const float norm = egr_sqrt_rn(egr_dot3(ld.x, ld.y, ld.z, ld.x, ld.y, ld.z)); // :41
    const float rr = egr_rcp_refined(norm);
    const float inv = egr_div_rn(1.0f, norm, rr);
    dhat.x = ld.x * inv, dhat.y = ld.y * inv, dhat.z = ld.z * inv;          // :42
    const float tl = egr_dot3(-lo.x, -lo.y, -lo.z, dhat.x, dhat.y, dhat.z); // :43
    t = egr_div_rn(tl, norm, rr);                                           // :44
    u.x = egr_madd(tl, dhat.x, lo.x), u.y = egr_madd(tl, dhat.y, lo.y), u.z = egr_madd(tl, dhat.z, lo.z); // :45
    behind = egr_dot3(lo.x, lo.y, lo.z, ld.x, ld.y, ld.z) > 0.0f;           // :36
    outside = egr_dot3(u.x, u.y, u.z, u.x, u.y, u.z) > 1.0f;                // :48-49
}
EGR_DI void candidate_geometry(const float4 &w0, const float4 &w1, const float4 &w2, const f3 &o, const f3 &d, f3 &lo, f3 &ld, f3 &dhat, float &t, f3 &u,
                               bool &behind, bool &outside) {
    lo = object_origin(w0, w1, w2, o);
    candidate_geometry_from(w0, w1, w2, lo, d, ld, dhat, t, u, behind, outside);
}
#pragma clang fp contract(fast)
// Wave-uniform node fetch: constant address space + uniform index => one s_load_dwordx8 through the scalar cache
// (the tree is read-only for the whole launch) instead of a 64-lane vector load.
typedef float egr_v8f __attribute__((ext_vector_type(8)));
EGR_DI void load_node_uniform(const float4 *nodes, uint32_t un, float4 &n0, float4 &n1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) egr_v8f *p = (const __attribute__((address_space(4))) egr_v8f *)nodes;
    const egr_v8f x = p[un];
    n0 = make_float4(x[0], x[1], x[2], x[3]);
    n1 = make_float4(x[4], x[5], x[6], x[7]);
#else
    n0 = nodes[2 * un], n1 = nodes[2 * un + 1];
#endif
}
typedef float egr_v4f __attribute__((ext_vector_type(4)));
EGR_DI float4 load_f4_uniform(const float4 *base, uint32_t idx) { // idx must be wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    const egr_v4f x = ((const __attribute__((address_space(4))) egr_v4f *)base)[idx];
    return make_float4(x[0], x[1], x[2], x[3]);
#else
    return base[idx];
#endif
}
EGR_DI uint4 load_u4_uniform(const uint4 *base, uint32_t idx) { // idx must be wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t egr_v4u __attribute__((ext_vector_type(4)));
    const egr_v4u x = ((const __attribute__((address_space(4))) egr_v4u *)base)[idx];
    return make_uint4(x[0], x[1], x[2], x[3]);
#else
    return base[idx];
#endif
}
// Segment [tmin,tmax] vs a quantised child box in the frame's cell coordinates: plane distance = fma(cell, invq, ncq)
// with invq = 1/(d*scale), ncq = -origin_cell*invq. Cells 0 / 65535 are the out-of-frame sentinels (-inf / +inf).
// The fused form moves a plane by < 0.01 cell (boxes carry a full extra cell each side); for an exactly axis-parallel
// ray both products are inf and the axis drops out (conservative).
// The interval part of the slab test: [max of the per-axis entries, min of the per-axis exits] clipped to [tmin, tmax].
EGR_DI bool slab_clip(float ax, float bx, float ay, float by, float az, float bz, float tmin, float tmax) {
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return t0 <= t1;
}
EGR_DI bool qslab_hit(uint4 q, f3 invq, f3 ncq, float tmin, float tmax) {
    const uint32_t lx = q.x & 0xFFFFu, ly = q.x >> 16, lz = q.y & 0xFFFFu, hx = q.y >> 16, hy = q.z & 0xFFFFu, hz = q.z >> 16;
    const float flx = lx == 0u ? -3.0e38f : (float)lx, fly = ly == 0u ? -3.0e38f : (float)ly, flz = lz == 0u ? -3.0e38f : (float)lz;
    const float fhx = hx == 65535u ? 3.0e38f : (float)hx, fhy = hy == 65535u ? 3.0e38f : (float)hy, fhz = hz == 65535u ? 3.0e38f : (float)hz;
    const float ax = fmaf(flx, invq.x, ncq.x), bx = fmaf(fhx, invq.x, ncq.x);
    const float ay = fmaf(fly, invq.y, ncq.y), by = fmaf(fhy, invq.y, ncq.y);
    const float az = fmaf(flz, invq.z, ncq.z), bz = fmaf(fhz, invq.z, ncq.z);
    return slab_clip(ax, bx, ay, by, az, bz, tmin, tmax);
}
// Same test when no box of the tree left the build frame (the common case; DeviceView::out_of_frame == 0): no sentinels.
EGR_DI bool qslab_hit_inframe(uint4 q, f3 invq, f3 ncq, float tmin, float tmax) {
    const float flx = (float)(q.x & 0xFFFFu), fly = (float)(q.x >> 16), flz = (float)(q.y & 0xFFFFu);
    const float fhx = (float)(q.y >> 16), fhy = (float)(q.z & 0xFFFFu), fhz = (float)(q.z >> 16);
    const float ax = fmaf(flx, invq.x, ncq.x), bx = fmaf(fhx, invq.x, ncq.x);
    const float ay = fmaf(fly, invq.y, ncq.y), by = fmaf(fhy, invq.y, ncq.y);
    const float az = fmaf(flz, invq.z, ncq.z), bz = fmaf(fhz, invq.z, ncq.z);
    return slab_clip(ax, bx, ay, by, az, bz, tmin, tmax);
}
EGR_DI float4 fetch_a2(const float4 *p) { return *p; }
EGR_DI float4 fetch_a2(float4 v) { return v; }
// OptiX's instance test restated: segment [tmin,tmax] of the object-space ray vs the unit cube.
EGR_DI bool hits_unit_cube(f3 lo, f3 ld, float tmin, float tmax) {
    f3 inv = mk3(__builtin_amdgcn_rcpf(ld.x), __builtin_amdgcn_rcpf(ld.y), __builtin_amdgcn_rcpf(ld.z)); // v_rcp_f32, 1 ulp
    float t0 = tmin, t1 = tmax;
    bool ok = true;
#define EGR_AXIS(c)                                                            \
    if (ld.c != 0.0f) {                                                        \
        float a = (-1.0f - lo.c) * inv.c, b = (1.0f - lo.c) * inv.c;           \
        t0 = fmaxf(t0, fminf(a, b));                                           \
        t1 = fminf(t1, fmaxf(a, b));                                           \
    } else if (lo.c < -1.0f || lo.c > 1.0f)                                    \
        ok = false;
    EGR_AXIS(x) EGR_AXIS(y) EGR_AXIS(z)
#undef EGR_AXIS
    return ok && t0 <= t1;
}

// The same test with IEEE divisions, operation for operation what the CPU oracle evaluates (exact-statistics build only).
EGR_DI bool hits_unit_cube_exact(f3 lo, f3 ld, float tmin, float tmax) {
    float t0 = tmin, t1 = tmax;
    bool ok = true;
#define EGR_AXIS(c)                                                            \
    if (ld.c != 0.0f) {                                                        \
        float a = (-1.0f - lo.c) / ld.c, b = (1.0f - lo.c) / ld.c;             \
        t0 = fmaxf(t0, fminf(a, b));                                           \
        t1 = fminf(t1, fmaxf(a, b));                                           \
    } else if (lo.c < -1.0f || lo.c > 1.0f)                                    \
        ok = false;
    EGR_AXIS(x) EGR_AXIS(y) EGR_AXIS(z)
#undef EGR_AXIS
    return ok && t0 <= t1;
}

// ---------------------------------------------------------------------------------------------------------
// forward chain: teams, the candidate test and the pair walk
// ---------------------------------------------------------------------------------------------------------
// A workgroup of the forward chain is a TEAM of EGR_TEAM waves. Every wave still owns its tiles alone (its own queue pulls, stack,
// leaf buffer, ray table, candidate scratch: waves never wait for each other while there are tiles), but the LDS is shared, and that
// is what lets SEVERAL WAVES WORK ON ONE HEAVY TILE: a wave that finds the task queue empty - or waits for the helpers of its own walk -
// looks for OFFERS, and a walking wave whose pair stack is long while a team mate looks for work puts every other pair of the lower part
// of it (up to EGR_BOX pairs) on offer. The taker walks those pairs on its own stack and leaf buffer against the OWNER's ray table, its
// accepted candidates take slots of the owner's lists through the owner's LDS counters - exactly what the owner would have done with
// them, in another order (the list order of a ray is an implementation matter) - and it may pass part of them on. A rank
// of an 8-way partition has one tile per wave slot and its launch lasts as long as its heaviest tile's chain. On by default
// (egr_team_help_on) - a ray's list order depends on timing, which no output sees except through the order of EXACT depth ties.
#ifndef EGR_TEAM
#define EGR_TEAM 16 // waves per workgroup of the forward chain's team build: the chain exists twice, as teams of this size for launches with
                    // egr_set_team_help(1) and as single-wave workgroups for all others (a team's LDS stays allocated until its last wave
                    // leaves, which costs a whole image 3 % of the chain when nobody helps; 16 = all the waves of a CU can help each other:
                    // rank 0 of an emulated 8-way partition, forward chain 1.60-1.65 ms against 1.66-1.87 with 4 and 1.79 without help)
#endif
#ifndef EGR_BOX
#define EGR_BOX 128 // (ray, node) pairs one offer holds
#endif
#ifndef EGR_DONATE_MIN
#define EGR_DONATE_MIN 96 // a walking wave offers half of its stack from this height on (48 ... 128 make no difference: sweeps r5k)
#endif
#define EGR_BOX_CLAIMED 0x80000000u
#define EGR_EXT_LOCKED (EGR_EXT_NONE - 2u) // a wave of the team is fetching this ray's extension block right now

// The waves of a team run independently, so the kernel has NO workgroup barrier. Within one wave LDS instructions execute in program
// order (a lane reads what another lane of its wave stored by an earlier instruction): all a wave needs between phases that exchange
// data through LDS is that the compiler keeps the order.
EGR_DI void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (volatile accesses keep their generic address space - the compiler would emit flat_load / flat_store for LDS words - so the cast names LDS)
typedef __attribute__((address_space(3))) uint32_t egr_lds_u32;
EGR_DI uint32_t lds_peek(const uint32_t *p) { return *(const volatile egr_lds_u32 *)p; }
EGR_DI void lds_poke(uint32_t *p, uint32_t x) { *(volatile egr_lds_u32 *)p = x; }
EGR_DI uint32_t uniform_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

struct FwdConst { // launch constants of a forward wave
    float exp_power, transmittance_threshold, backfacing_max_dist, backfacing_thr, far_plane;
    int num_bounces;
    bool sentinels;
    const float4 *__restrict__ app;
    const uint4 *__restrict__ wnodes;
};
struct WalkShared { // one wave's LDS
    uint32_t pstk[EGR_PSTK];  // (ray << 26 | wide node); entries beyond EGR_PSTK spill to the wave's global column
    uint32_t lbuf[EGR_LBUF];  // (ray << 26 | record) awaiting evaluation; a walk batch adds <= 8 GPOP x 8 and only runs when that fits
    float4 rayp[EGR_WAVE][3]; // per ray: (o.xyz, d.x) (d.yz, invq.xy) (invq.z, ncq.xyz)
    uint32_t gcnt[EGR_WAVE], gtrav[EGR_WAVE]; // accepted / counted candidates per ray
    uint32_t gext[EGR_WAVE];  // extension block of a ray's candidate list (EGR_EXT_NONE: none)
    uint32_t wc[4 * EGR_NSTEPS]; // this wave's ray / candidate / composited / accepted counts per step
};
template <int TEAM> struct TeamShared {
    uint32_t done;                 // waves of the team that found the task queue empty (they only help from then on)
    uint32_t hungry;               // waves looking for an offer right now (out of tiles, or waiting for the helpers of their own walk)
    uint32_t box_count[TEAM];  // the offer of wave w: pairs | rays' owner << 16 (0: none; EGR_BOX_CLAIMED set: being copied out)
    uint32_t busy[TEAM];       // other waves that hold pairs of wave w's current walk (a wave announces itself here BEFORE it claims an offer)
    uint32_t ctx[TEAM][5];     // the walk wave w's offers belong to: step | seg << 8, seg_lo, seg_hi, near plane, the wave whose rays these are
    uint32_t box[TEAM][EGR_BOX];
};
#if defined(EGR_TRAVERSAL_STATS) || defined(EGR_TASK_TIMES)
#define EGR_WALK_STATS 1
#endif
struct WalkStats { // diagnostic builds (EGR_TRAVERSAL_STATS, EGR_TASK_TIMES)
    uint32_t visits = 0, leafhits = 0, inner = 0, outer = 0;
    uint32_t offers = 0, tall = 0; // team: offers made, walk batches that left the stack at EGR_DONATE_MIN pairs or more
};

// R2 for one (ray, gaussian) pair given the gaussian's W rows and live (.., opacity, sigma) record (shaders.cu:9-75).
// `prim` is the gaussian's SORTED POSITION (record index), not its id.
// returns 0: not this segment's / not met by the reference, 1: counted and rejected, 2: accepted (t, alpha valid), 3: response point outside
// the ellipsoid (rejected; counted only by the exact-statistics build). What a DEFAULT launch counts (num_traversed_per_pixel,
// egr_counters.candidates) is a property of the ray and the gaussian alone, not of how the tree was walked: the candidates whose response
// point lies INSIDE the gaussian's ellipsoid on the ray's segment - a subset of the reference's intersection-program invocations
// (every instance whose CUBE the segment overlaps, shaders.cu:33), which the exact-statistics build counts (egr_set_exact_stats).
// SEG0: the caller's copy of the loop only runs for segment 0 (the hot case: no runtime segment dispatch).
template <bool CUBE, bool SEG0, class A2>
EGR_DI int test_candidate(const FwdConst &fc, int step, int seg, float near_plane, const f3 &o, const f3 &d, uint32_t prim, const float4 &w0, const float4 &w1,
                          const float4 &w2, A2 a2src, float &t, float &alpha, const f3 *lo_pre = nullptr) {
    const float far_plane = fc.far_plane;
    f3 lo, ld, dhat, u;
    bool behind, outside;
    if (lo_pre != nullptr) { // (primary tiles: the object-space origin was computed once per leaf - the same expression, object_origin)
        lo = *lo_pre;
        candidate_geometry_from(w0, w1, w2, lo, d, ld, dhat, t, u, behind, outside);
    } else {
        candidate_geometry(w0, w1, w2, o, d, lo, ld, dhat, t, u, behind, outside); // :19-20, :36, :41-51
    }
    if constexpr (CUBE) {
        // exact-statistics build: the tree bounds the instance cubes, one walk over [tmin,tmax] meets every instance OptiX
        // would invoke the intersection program for; the cube test is the oracle's (IEEE division)
        if (!hits_unit_cube_exact(lo, ld, near_plane, far_plane)) return 0;
    } else {
        // each candidate is owned by exactly one of the three walked segments (forward_task.inc): counted / accepted once
        if constexpr (SEG0) {
            if (!(t >= near_plane && t <= far_plane)) return 0;
        } else if (!(seg == 0 ? (t >= near_plane && t <= far_plane) : seg == 1 ? (t < near_plane) : (t > far_plane))) return 0;
        // OptiX only invokes the intersection program when the instance's unit cube overlaps [tmin,tmax]. A response point
        // inside the unit sphere AND on the segment (segment 0) is itself a point of cube and segment, so the cube test is
        // implied for everything segment 0 ACCEPTS; of the hits it rejects, those behind the origin or back-facing are counted without
        // asking whether the reference would have looked at them, those outside the ellipsoid are not counted at all (default launches
        // count the candidates INSIDE their ellipsoid - include/egr_raytracer.h, egr_set_exact_stats -, the exact-statistics build counts
        // the reference's invocations). Segments 1 / 2 hold the Q1 hits, accepted only if the cube overlaps [tmin,tmax].
        if constexpr (!SEG0)
            if (seg != 0 && !hits_unit_cube(lo, ld, near_plane, far_plane)) return 0;
    }
    if (outside) return CUBE ? 1 : 3;               // :48-51 (the exact-statistics build counts every invocation, shaders.cu:33)
    if (behind) return 1;                           // :36
    if (step != 0 && t < fc.backfacing_max_dist) {  // :54-61 (world normal . object dir)
        const float4 n0 = fc.app[2 * prim], n1 = fc.app[2 * prim + 1]; // raw normal, record order (k_live)
        f3 gn = mk3(n0.w, n1.x, n1.y);
        if (sqrtf(egr_dot3(gn.x, gn.y, gn.z, gn.x, gn.y, gn.z)) > fc.backfacing_thr && egr_dot3(gn.x, gn.y, gn.z, dhat.x, dhat.y, dhat.z) > 0.0f) return 1;
    }
    const float4 a2 = fetch_a2(a2src);                        // live quarter of the record, only now
    f3 x = u * a2.w;                                          // :64
    float gaussval = egr_gaussian_sq(egr_dot3(x.x, x.y, x.z, x.x, x.y, x.z), fc.exp_power); // :65 (spelled-out fusion: the owner's and a helper's copy of this test agree to the bit)
    alpha = EGR_MAX_ALPHA * gaussval * a2.z;                   // kernel.cu:14-16
    return 2;
}

// One evaluation batch worked off: lane = one (ray, gaussian) pair whose 64-B record (w0..w2 = rows of W, w3 = live quarter) has arrived. The candidate
// test, then an accepted candidate takes a slot of its ray's list through the ray's LDS counter (the run in the wave's scratch, beyond cand_cap ONE
// extension block per ray). Shared by the pair walk of bounce tiles and by the primary tiles' leaf evaluation (forward_task.inc).
template <bool CUBE, bool SEG0>
EGR_DI void pair_eval_append(const DeviceView &v, const FwdConst &fc, WalkShared &rays, const size_t scratch0, const int step, const int seg, const float near_plane, const bool have,
                             const uint32_t er, const uint32_t pidx, const float4 &w0, const float4 &w1, const float4 &w2, const float4 &w3, bool &g_over, WalkStats &st) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    (void)lane;
    int res = 0;
    float t = 0.0f, alpha = 0.0f;
    if (have) {
        const float4 q0 = rays.rayp[er][0], q1 = rays.rayp[er][1];
        res = test_candidate<CUBE, SEG0>(fc, step, seg, near_plane, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), pidx, w0, w1, w2, w3, t, alpha);
    }
#ifdef EGR_WALK_STATS
    st.leafhits += have ? 1u : 0u;
    st.outer += (lane == 0);
#endif
    if (res == 1 || res == 2) atomicAdd(&rays.gtrav[er], 1u);
    uint32_t at = 0u;
    if (res == 2) at = atomicAdd(&rays.gcnt[er], 1u);
    const bool in_ext = res == 2 && at >= v.cand_cap;
    if (res == 2 && !in_ext) {
        const size_t slot = (scratch0 + er) * v.cand_cap + at;
        v.cand_keys[scratch0 * v.cand_cap + ((((size_t)(at >> 2) * EGR_WAVE + er) << 2) + (at & 3u))] = t; // (forward_decl.inc: EGR_KEY_AT of the owner's region)
        v.cand_vals[slot] = make_float2(alpha, u2f(pidx)); // (a pair walk's lists are contiguous runs: forward_decl.inc)
    }
    if (__ballot(in_ext) != 0ull) { // rare: some list outgrew its run - it continues in ONE extension block per ray
        for (;;) { // rays that need a block and have none, one after the other (wave-uniform loop)
            const unsigned long long pend = __ballot(in_ext && lds_peek(&rays.gext[er]) == EGR_EXT_NONE);
            if (pend == 0ull) break;
            const int src = __ffsll((long long)pend) - 1;
            if (lane == src && atomicCAS(&rays.gext[er], EGR_EXT_NONE, EGR_EXT_LOCKED) == EGR_EXT_NONE) { // (a team mate working on the same ray may be faster)
                const uint32_t e = atomicAdd(v.control + CW_EXT_BUMP, 1u);
                lds_poke(&rays.gext[er], e < v.ext_blocks_cap ? e : EGR_EXT_NONE - 1u); // EGR_EXT_NONE - 1 = none left
            }
            wave_sync();
        }
        if (in_ext) {
            uint32_t e = lds_peek(&rays.gext[er]);
            while (e == EGR_EXT_LOCKED) __builtin_amdgcn_s_sleep(1), e = lds_peek(&rays.gext[er]);
            const uint32_t k = at - v.cand_cap;
            if (e < EGR_EXT_NONE - 1u && k < EGR_EXT_BLOCK) {
                v.ext_keys[(size_t)e * EGR_EXT_BLOCK + k] = t;
                v.ext_vals[(size_t)e * EGR_EXT_BLOCK + k] = make_float2(alpha, u2f(pidx));
            } else {
                g_over = true;
            }
        }
    }
}

// ---- pair walk. Work items are PAIRS, not rays: a wave-wide LIFO of (ray, node) pairs and a buffer of (ray, leaf) pairs,
// both in LDS. A walk iteration pops up to 8 x EGR_GPOP pairs; lane m of group g tests child slot m of the g-th popped node
// (one 16-B load per lane = the node's 128-B line per group), hits are compacted with wave-wide ballots: inner children
// back onto the stack, leaves into the leaf buffer. Whenever 64 leaf pairs wait, ONE LANE PER PAIR evaluates them (every
// lane busy, no per-ray open / close, no leaf queue in global memory). Accepted candidates take a slot of their ray's list
// through an LDS counter; the total transmittance is multiplied up from the finished list (its order, like the
// reference's insertion order, is an implementation matter: deterministic for a wave that walks alone).
// The loop exists once per decode of the child boxes (SENT: out-of-frame sentinels or not, wave-uniform for the whole launch) and once more
// for segment 0 without sentinels (SEG0; nearly every walk of a bounce step) - the choice is made once per walk instead of once per slot batch.
// `mine`: stack and leaf buffer of the executing wave; `rays`: ray table, per-ray counters and (through scratch0) candidate lists of the
// wave that OWNS the tile - the same wave unless the pairs were taken from an offer (team_help). The caller enters with `top` pairs on mine.pstk.
template <bool SENT, bool SEG0, bool CUBE, int TEAM>
EGR_DI void pair_walk(const DeviceView &v, const FwdConst &fc, WalkShared &mine, WalkShared &rays, TeamShared<TEAM> &team, const int self, const size_t scratch0,
                      uint32_t *__restrict__ gstk, const int step, const int seg, const float seg_lo, const float seg_hi, const float near_plane, uint32_t top,
                      bool &g_over, WalkStats &st) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    const uint32_t m = (uint32_t)lane & 7u, grp = (uint32_t)lane >> 3;
    constexpr uint32_t GSTK_CAP = (uint32_t)EGR_GSTK * EGR_WAVE;
    const uint4 *__restrict__ wnodes = fc.wnodes;
    uint32_t nl = 0u; // wave-uniform height of the leaf buffer (top: of the pair stack)
    uint32_t hungry_seen = 0u;
    for (;;) {
        // One iteration = (up to) one evaluation batch AND one walk batch: the record fetches of the batch of 64 leaf pairs and
        // the node fetches of the popped pairs are issued together, then both are worked off - a tile's critical path is its
        // chain of dependent fetches (every batch needs what the previous one found), so two kinds of work per round trip
        // instead of one shorten it (the walk refills the leaf buffer with ~40 pairs per iteration).
        const bool do_eval = nl >= (uint32_t)EGR_WAVE || (top == 0u && nl > 0u); // wave-uniform
        if (!do_eval && top == 0u) {
            if constexpr (TEAM > 1) {
                // an offer of this wave that nobody took comes back. (One that is being copied out right now is the taker's: it announced
                // itself in busy[rays' owner] before it claimed the offer, and that owner waits for it - forward_task.inc.)
                uint32_t r = 0u;
                if (lane == 0) {
                    const uint32_t c = lds_peek(&team.box_count[self]);
                    if (c != 0u && !(c & EGR_BOX_CLAIMED) && atomicCAS(&team.box_count[self], c, c | EGR_BOX_CLAIMED) == c) r = c;
                }
                r = uniform_u32(r);
                if (r != 0u) {
                    r &= 0xFFFFu;
                    for (uint32_t i = (uint32_t)lane; i < r; i += EGR_WAVE) mine.pstk[i] = team.box[self][i];
                    top = r;
                    wave_sync();
                    if (lane == 0) lds_poke(&team.box_count[self], 0u);
                    continue;
                }
            }
            break;
        }
        // ---------------- issue: evaluation batch (64 (ray, leaf) pairs, one lane each)
        bool have = false;
        uint32_t er = 0u, pidx = 0u;
        float4 w0 = make_float4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0; // the whole 64-B record: rows of W + the live quarter (same line)
        if (do_eval) {
            const uint32_t take = min(nl, (uint32_t)EGR_WAVE);
            nl -= take;
            have = (uint32_t)lane < take;
            const uint32_t pr = mine.lbuf[nl + min((uint32_t)lane, take - 1u)];
            er = pr >> 26, pidx = pr & 0x03FFFFFFu;
            if (have) w0 = v.inst_w[4 * pidx], w1 = v.inst_w[4 * pidx + 1], w2 = v.inst_w[4 * pidx + 2], w3 = v.inst_w[4 * pidx + 3];
        }
        // ---------------- issue: walk batch (pop up to 8 x EGR_GPOP (ray, node) pairs)
        // (a walk batch only runs when the leaf pairs it can add fit the buffer: otherwise this iteration only evaluates)
        const uint32_t npop = nl + (uint32_t)EGR_WALK_MAX_LEAVES <= (uint32_t)EGR_LBUF ? min(top, 8u * (uint32_t)EGR_GPOP) : 0u;
        uint32_t pw[EGR_GPOP];
        uint4 sl_[EGR_GPOP];
        if constexpr (TEAM > 1) hungry_seen = lds_peek(&team.hungry); // (read with the stack: one LDS round trip)
        if (npop != 0u) {
#pragma unroll
            for (int u = 0; u < EGR_GPOP; u++) pw[u] = mine.pstk[min(top - 1u - min(grp + 8u * (uint32_t)u, top - 1u), (uint32_t)EGR_PSTK - 1u)];
            if (top > (uint32_t)EGR_PSTK) { // some popped pairs live in the global spill column (rare; volatile keeps it a global load)
#pragma unroll
                for (int u = 0; u < EGR_GPOP; u++) {
                    const uint32_t e = top - 1u - min(grp + 8u * (uint32_t)u, top - 1u);
                    if (grp + 8u * (uint32_t)u < npop && e >= (uint32_t)EGR_PSTK) pw[u] = *reinterpret_cast<const volatile uint32_t *>(gstk + (e - EGR_PSTK));
                }
            }
#pragma unroll
            for (int u = 0; u < EGR_GPOP; u++) {
                // (a group beyond the popped pairs re-reads the bottom pair's node - a valid entry, the index above is clamped - and is
                // turned into an empty slot: no exec-mask detour around the load)
                sl_[u] = wnodes[(size_t)(pw[u] & 0x03FFFFFFu) * EGR_WIDTH + m];
                if (!(grp + 8u * (uint32_t)u < npop)) sl_[u].w = EGR_EMPTY_SLOT;
            }
            top -= npop;
#ifdef EGR_WALK_STATS
            st.visits += (m == 0u) ? min(npop > grp ? (npop - grp + 7u) / 8u : 0u, (uint32_t)EGR_GPOP) : 0u;
            st.inner += (lane == 0);
#endif
        }
        wave_sync(); // the reads of both buffers above come before the pushes below
        // ---------------- work off: evaluation batch
        if (do_eval) pair_eval_append<CUBE, SEG0>(v, fc, rays, scratch0, step, seg, near_plane, have, er, pidx, w0, w1, w2, w3, g_over, st);
        // ---------------- work off: walk batch
        if (npop != 0u) {
#pragma unroll
            for (int u = 0; u < EGR_GPOP; u++) {
                const uint32_t rtag = pw[u] & 0xFC000000u; // the pair's ray, in place
                const float4 q1 = rays.rayp[pw[u] >> 26][1], q2 = rays.rayp[pw[u] >> 26][2];
                const uint4 sl = sl_[u];
                // (masks built on the scalar side: one ballot of the box test, one of the leaf bit; an unpopped slot is an empty slot)
                const bool hit = (SENT ? qslab_hit(sl, mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), seg_lo, seg_hi) : qslab_hit_inframe(sl, mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), seg_lo, seg_hi)) & (sl.w != EGR_EMPTY_SLOT);
                const bool leaf = (int)sl.w < 0;
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit), lm = __builtin_amdgcn_ballot_w64(leaf);
                const unsigned long long im = hm & ~lm, fm = hm & lm;
                const uint32_t at = top + __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                if (top + (uint32_t)EGR_WAVE <= (uint32_t)EGR_PSTK) { // wave-uniform: every push of this batch stays in the LDS part (the usual case)
                    if (hit & !leaf) mine.pstk[at] = rtag | sl.w;
                    top += (uint32_t)__popcll(im);
                } else {
                    if (hit & !leaf) {
                        if (at < (uint32_t)EGR_PSTK) mine.pstk[at] = rtag | sl.w;
                        else if (at - EGR_PSTK < GSTK_CAP) gstk[at - EGR_PSTK] = rtag | sl.w;
                        else g_over = true;
                    }
                    top = min(top + (uint32_t)__popcll(im), (uint32_t)EGR_PSTK + GSTK_CAP);
                }
                if (hit & leaf) mine.lbuf[nl + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = rtag | (sl.w & ~EGR_LEAF_FLAG);
                nl += (uint32_t)__popcll(fm);
            }
        }
        wave_sync();
#ifdef EGR_WALK_STATS
        st.tall += (lane == 0 && top >= (uint32_t)EGR_DONATE_MIN) ? 1u : 0u;
#endif
        if constexpr (TEAM > 1) {
            // a team mate looks for work and this stack is long: its lower half goes on offer
            if (uniform_u32(hungry_seen) != 0u && top >= (uint32_t)EGR_DONATE_MIN && top <= (uint32_t)EGR_PSTK && uniform_u32(lds_peek(&team.box_count[self])) == 0u) {
                // EVERY OTHER pair of the lower part of the stack goes, the rest moves down. Both halves then hold subtrees of every size (the
                // bottom holds the pairs nearest the root, i.e. the largest subtrees still to walk, the top the crumbs of the subtree the wave
                // is in). Measured on rank 0 of an emulated 8-way partition: offers from the top shortened the heaviest walk by a quarter at
                // best; offers of the bottom half left the owner waiting for its helpers for as long as its own half had taken.
                const uint32_t give = min(top >> 1, (uint32_t)EGR_BOX);
                for (uint32_t base = 0u; base < top; base += EGR_WAVE) {
                    const uint32_t i = base + (uint32_t)lane;
                    const uint32_t x = mine.pstk[min(i, (uint32_t)EGR_PSTK - 1u)];
                    wave_sync();
                    if (i < 2u * give) {
                        if (i & 1u) team.box[self][i >> 1] = x;
                        else mine.pstk[i >> 1] = x;
                    } else if (i < top) mine.pstk[i - give] = x;
                }
                top -= give;
                wave_sync();
                if (lane == 0) lds_poke(&team.box_count[self], give | (lds_peek(&team.ctx[self][4]) << 16));
#ifdef EGR_WALK_STATS
                st.offers += (lane == 0);
#endif
            }
        }
    }
}

// A turn of a wave that looks for work: take an offer of a team mate and walk it - on this wave's own stack and leaf buffer, against the
// ray table, per-ray counters and candidate lists of the wave whose rays the pairs belong to (ONE generic instance of pair_walk:
// sentinel-aware, any segment). Returns false when there was no offer. The caller counts itself in team.hungry while it looks.
template <bool CUBE, int TEAM> EGR_DI bool team_help(const DeviceView &v, const FwdConst &fc, WalkShared *wsh, TeamShared<TEAM> &team, const int wv, const uint32_t slot0, bool &g_over) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    for (int dlt = 1; dlt < TEAM; dlt++) {
        const int w = (wv + dlt) % TEAM;
        uint32_t k = 0u, o = 0u;
        if (lane == 0) {
            const uint32_t c = lds_peek(&team.box_count[w]);
            if (c != 0u && !(c & EGR_BOX_CLAIMED)) {
                o = (c >> 16) & 0xFFu;
                atomicAdd(&team.busy[o], 1u); // first: the rays' owner must never see "nobody busy" while these pairs are open
                if (atomicCAS(&team.box_count[w], c, c | EGR_BOX_CLAIMED) == c) k = c & 0xFFFFu; // (ctx[w] cannot change while an offer of w is open)
                else atomicSub(&team.busy[o], 1u);
            }
        }
        k = uniform_u32(k);
        if (k == 0u) continue;
        const int owner = (int)uniform_u32(o);
        if (lane == 0) atomicSub(&team.hungry, 1u);
        WalkShared &mine = wsh[wv];
        for (uint32_t i = (uint32_t)lane; i < k; i += EGR_WAVE) mine.pstk[i] = team.box[w][i];
        const uint32_t c0 = uniform_u32(lds_peek(&team.ctx[w][0])), c1 = uniform_u32(lds_peek(&team.ctx[w][1])), c2 = uniform_u32(lds_peek(&team.ctx[w][2])), c3 = uniform_u32(lds_peek(&team.ctx[w][3]));
        // this wave's own offers (it may pass on part of what it took) describe the same walk; its last offer may still be on its way out
        while (uniform_u32(lds_peek(&team.box_count[wv])) != 0u) __builtin_amdgcn_s_sleep(1);
        if (lane == 0) lds_poke(&team.ctx[wv][0], c0), lds_poke(&team.ctx[wv][1], c1), lds_poke(&team.ctx[wv][2], c2), lds_poke(&team.ctx[wv][3], c3), lds_poke(&team.ctx[wv][4], (uint32_t)owner);
        wave_sync();
        if (lane == 0) lds_poke(&team.box_count[w], 0u); // w may offer the next batch
        const int step = (int)(c0 & 0xFFu), seg = (int)(c0 >> 8);
        const size_t scratch0 = ((size_t)slot0 + (size_t)owner) * EGR_WAVE; // the OWNER's candidate lists
        uint32_t *gstk = v.stack_spill + ((size_t)slot0 + (size_t)wv) * EGR_GSTK * EGR_WAVE; // this wave's own spill column
        WalkStats st;
        pair_walk<true, false, CUBE, TEAM>(v, fc, mine, wsh[owner], team, wv, scratch0, gstk, step, seg, u2f(c1), u2f(c2), u2f(c3), k, g_over, st);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the list entries written for the owner
#ifdef EGR_WALK_STATS
        if (lane == 0) atomicAdd(v.control + CW_DBG3 + 13, 1u), atomicAdd(v.control + CW_DBG3 + 14, st.inner);
#endif
        if (lane == 0) atomicSub(&team.busy[owner], 1u), atomicAdd(&team.hungry, 1u);
        return true;
    }
    return false;
}
// ... for as long as `wanted()` says (wave-uniform): the waiting loop of an owner whose walk still has helpers, and of a wave without tiles.
template <bool CUBE, int TEAM, class F> EGR_DI void team_help_while(const DeviceView &v, const FwdConst &fc, WalkShared *wsh, TeamShared<TEAM> &team, const int wv, const uint32_t slot0, bool &g_over, F wanted) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    if (!wanted()) return;
    if (lane == 0) atomicAdd(&team.hungry, 1u);
    while (wanted())
        if (!team_help<CUBE, TEAM>(v, fc, wsh, team, wv, slot0, g_over)) __builtin_amdgcn_s_sleep(4);
    if (lane == 0) atomicSub(&team.hungry, 1u);
}

// ---------------------------------------------------------------------------------------------------------
// per-launch prologue: Raytracer::raytrace host part (raytracer.cpp:82-86), on the device, no host sync
// ---------------------------------------------------------------------------------------------------------
__global__ void k_prologue(DeviceView v, int grads) {
    int t = threadIdx.x;
    if (t < CW_RESET_END) v.control[t] = 0;
    for (int w = CW_DBG + t; w < CW_COUNT; w += (int)blockDim.x) v.control[w] = 0; // (diagnostic words: per launch)
    if (t < 16) v.control[CW_DBG2 + t] = 0;
    for (uint32_t q = t; q < EGR_QUEUE_WORDS * v.num_strands; q += blockDim.x) v.queues[q] = 0;
    if (t < 12) v.control[CW_DBG3 + t] = ((t & 3) < 2) ? 0xFFFFFFFFu : 0u;
    if (t == 0) {
        *v.meta.grads_enabled = grads ? 1 : 0;   // metadata.h:29
        *v.meta.total_num_calls += 1;             // metadata.h:30
    }
}
__global__ void k_epilogue(DeviceView v, int grads) {
    // raytracer.cpp:91-93 (the reference adds 1 whenever accumulate_samples is set)
    if (threadIdx.x == 0) {
        if (*v.cfg.accumulate_samples) *v.fb.accumulated_sample_count += 1;
        unsigned long long rays = 0;
        for (int s = 0; s < EGR_NSTEPS; s++) rays += *reinterpret_cast<unsigned long long *>(v.control + CW_RAYS + 2 * s);
        *reinterpret_cast<unsigned long long *>(v.control + CW_LIFE_RAYS) += rays;
        v.control[CW_LIFE_LAUNCHES] += 1;
    }
}
// per-launch live record: activated appearance + (opacity, sigma). Reads the CURRENT parameter tensors, like
// the reference's read_* helpers do inside the launch (utils/helpers.cu:10-33), while the transforms in inst_w / inst_m stay
// snapshots of the last update_bvh / rebuild_bvh. Grad launches also refresh the two live quantities of the backward record:
// exp(scale) and the raw quaternion, which the reference's backward reads from the parameter tensors (backward_pass.cu:68-70)
// next to OptiX's snapshot transforms (:75-78).
__global__ void __launch_bounds__(256) k_live(DeviceView v, int grads) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= v.n) return;
    const egr_gaussians &g = v.g;
    float o = sigmoid_act(g.opacity[i]);
    float sigma = compute_scaling_factor(o, *v.cfg.alpha_threshold, *v.cfg.exp_power);
    const size_t pos = v.pos_of_gid[i]; // stored at the sorted position
    float4 *app = v.app + 2 * pos;
    app[0] = make_float4(relu_act(g.rgb[3 * i]), relu_act(g.rgb[3 * i + 1]), relu_act(g.rgb[3 * i + 2]), g.normal[3 * i]);
    app[1] = make_float4(g.normal[3 * i + 1], g.normal[3 * i + 2], clip01_act(g.f0[3 * i]), clip01_act(g.f0[3 * i + 1]));
    // the 4th quarter of the 64-B test record (rows of W are the first three): what a candidate test needs besides W
    v.inst_w[4 * pos + 3] = make_float4(clip01_act(g.f0[3 * i + 2]), clip01_act(g.roughness[i]), o, sigma);
    if (grads) {
        float4 *im = v.inst_m + 4 * pos;
        im[0].w = expf(g.scale[3 * i]), im[1].w = expf(g.scale[3 * i + 1]), im[2].w = expf(g.scale[3 * i + 2]);
        im[3] = reinterpret_cast<const float4 *>(g.rotation)[i];
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward: all bounce steps of every tile of this rank
// ---------------------------------------------------------------------------------------------------------
// The fused per-tile chain: a wave takes a tile through ALL its steps (trace, step epilogue, trace, ...) before it takes the
// next tile. A rank's forward time is then max over tiles of (sum of its steps) instead of the sum over steps of (max over
// tiles): with one kernel per step every kernel ends with a tail as long as its heaviest tile, and a tile's cost at one step
// says nothing about its cost at the next (measured correlation -0.2 .. -0.07; round 1: 315 -> 364 Mrays/s).
// CUBE = the exact-statistics build (egr_set_exact_stats): the tree bounds the reference's instance CUBES and every cube overlap
// is counted, so num_traversed_per_pixel is the reference's number (see forward_task.inc); images are the same.
#ifndef EGR_ARENA_CHUNK
#define EGR_ARENA_CHUNK 8 // hit-arena blocks a wave takes from the bump counter at a time (up to CHUNK - 1 blocks per resident wave stay unused: 29 k of the >= 469 k blocks of the reference's default capacity)
#endif
#ifndef EGR_FWD_WAVES
#define EGR_FWD_WAVES 4 // waves per SIMD the forward chain is built for (register budget 512 / EGR_FWD_WAVES)
#endif
template <bool GRADS, bool CUBE, int TEAM> __global__ void __launch_bounds__(EGR_WAVE * TEAM) __attribute__((amdgpu_waves_per_eu(EGR_FWD_WAVES, EGR_FWD_WAVES))) k_forward_chain(DeviceView v) {
#include "forward_decl.inc"
    if (lane < 4 * EGR_NSTEPS) wc[lane] = 0u;
    if (threadIdx.x == 0) team.done = 0u, team.hungry = 0u;
    if (threadIdx.x < TEAM) team.box_count[threadIdx.x] = 0u, team.busy[threadIdx.x] = 0u;
    __syncthreads(); // the kernel's only workgroup barrier: from here on the waves of a team run independently
    uint32_t cur_q = blockIdx.x & 7u;
    uint32_t arena_next = 0u, arena_end = 0u; // this wave's run of hit-arena blocks (forward_task.inc)

    for (;;) {
        const uint32_t tq = slot < v.num_slots ? wave_next_task(v.queues, v.task_count, cur_q, lane) : 0xFFFFFFFFu;
        if (tq == 0xFFFFFFFFu) break;
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 9 // diagnostic build: stamps of the WHOLE chain of a task (start, end of every step) in its first pixels
        unsigned long long chain_t[EGR_NSTEPS + 1] = {__builtin_amdgcn_s_memrealtime(), 0ull, 0ull, 0ull};
        uint32_t chain_leaves = 0u;
#endif
#ifdef EGR_TRAVERSAL_STATS
        const unsigned long long tchain0 = __builtin_amdgcn_s_memtime();
        unsigned long long tepi = 0ull;
#endif
        uint32_t bwd_cost = 0u; // (grad launches) what this tile's backward will cost, roughly in microseconds: 8 per primary hit row, 8 per 64 bounce hits + 2 per bounce hit row
        for (int step = 0; step <= num_bounces; step++) {
            do { // (a `continue` in the step body ends the step)
                const float near_plane = step == 0 ? *v.cam.znear : 0.0f; // forward_pass.cu:8-11
#include "forward_task.inc"
                const uint32_t a = wave_sum_u32(active ? 1u : 0u), b = wave_sum_u32(active ? traversed : 0u), c2 = wave_sum_u32(active ? nhits : 0u);
                const uint32_t d2 = wave_sum_u32(active ? cnt : 0u);
                if (lane == 0) wc[4 * step] += a, wc[4 * step + 1] += b, wc[4 * step + 2] += c2, wc[4 * step + 3] += d2;
                if (GRADS) {
                    const uint32_t rows = wave_max_u32(active ? nhits : 0u);
                    bwd_cost += step == 0 ? 8u * rows : c2 / 8u + 2u * rows;
                }
            } while (false);
            // R4 / R5 of this step for the tile's rays
            const uint32_t etask = v.task_begin + tq;
            const TaskGeom etg = task_geom(v, etask, lane);
#ifdef EGR_TRAVERSAL_STATS
            const unsigned long long tepi0 = __builtin_amdgcn_s_memtime();
#endif
            if (etg.inside) step_epilogue_lane(v, step, GRADS, num_bounces, etg, state_of(v, etask, lane));
#ifdef EGR_TRAVERSAL_STATS
            tepi += __builtin_amdgcn_s_memtime() - tepi0;
#endif
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 9
            chain_t[step + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        if (GRADS && lane == 0) v.task_cost[v.task_begin + tq] = bwd_cost;
#ifdef EGR_TRAVERSAL_STATS
        if (lane == 0) { // CW_DBG2 + 8: step epilogues, + 12: whole chains (task pull to task end)
            atomicAdd(reinterpret_cast<unsigned long long *>(v.control + CW_DBG2 + 8), tepi);
            atomicAdd(reinterpret_cast<unsigned long long *>(v.control + CW_DBG2 + 12), __builtin_amdgcn_s_memtime() - tchain0);
        }
#endif
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 9
        {
            const TaskGeom ctg = task_geom(v, v.task_begin + tq, lane);
            if (lane <= EGR_NSTEPS && ctg.inside) v.stats.num_traversed_per_pixel[ctg.pixel_id] = (int32_t)(chain_t[lane] & 0x7FFFFFFFull);
            if (lane == 4 && ctg.inside) v.stats.num_traversed_per_pixel[ctg.pixel_id] = (int32_t)chain_leaves;
        }
#endif
    }
    wave_sync();
    if (lane < EGR_NSTEPS) {
        add64(v.control, CW_RAYS + 2 * lane, wc[4 * lane]), add64(v.control, CW_CAND + 2 * lane, wc[4 * lane + 1]), add64(v.control, CW_COMP + 2 * lane, wc[4 * lane + 2]);
        add64(v.control, CW_ACCEPTED + 2 * lane, wc[4 * lane + 3]);
    }
    if (TEAM > 1 && v.team_help) {
        // no tiles left for this wave: it helps its team mates with the walks of theirs until all of them are through
        if (lane == 0) atomicAdd(&team.done, 1u);
        bool h_over = false;
        team_help_while<CUBE, TEAM>(v, fc, wsh_all, team, wv, blockIdx.x * (uint32_t)TEAM, h_over, [&]() { return uniform_u32(lds_peek(&team.done)) < (uint32_t)TEAM; });
        if (h_over) atomicOr(v.control + CW_STATUS, EGR_STATUS_CANDIDATE_OVERFLOW);
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward: walked newest (farthest) hit first  (backward_pass.cu:3-222)
// ---------------------------------------------------------------------------------------------------------
// Gradient pre-reduction (cdna guide, Guideline 12): the 64 primary rays of a tile composite the same few dozen Gaussians,
// so per-hit contributions are first summed in a per-wave LDS hash table (ds_add_f32, open addressing on the
// record index) and flushed once per tile: ~8x fewer global atomics than backward_pass.cu:210-220's 15/22 per hit.
#ifndef EGR_GT_SLOTS
#define EGR_GT_SLOTS 80 // slots of the primary step's LDS table (any count >= 64; multiply-shift hash). Since a full table is flushed and refilled
                        // (round 4) a smaller one is as good: 64 / 72 / 80 / 94 slots -> 3.02 / 3.05 / 3.01-3.03 / 3.11 ms backward chain
                        // dense-init (same box). Before that: 94 beat 64 (hits without a slot left as records of their own), 128 and 256 cost waves per CU
#endif
#define EGR_GT_COMPS 22
#define EGR_GT_STRIDE 23 // floats per table slot ([slot][component], odd stride: lanes on different slots fall on different LDS banks)
#define EGR_GT_EMPTY 0xFFFFFFFFu
// component order of the LDS table, of a wide-add record (first 15) and of a gradient row (DeviceView::grad_rows)
enum : int { GC_OPA = 0, GC_SCALE = 1, GC_MEAN = 4, GC_ROT = 7, GC_RGB = 11, GC_WEIGHT = 14, GC_NORMAL = 15, GC_F0 = 18, GC_ROUGH = 21 };
#define EGR_ROW_STRIDE 32 // floats per gradient row: one 128-B line per gaussian
#define EGR_BWD_SYNC() wave_sync() // the backward chain's waves run independently (like the forward's): phases that exchange data through LDS within ONE wave need no more than program order

// SIXTEEN LANES PER RECORD add the 15 values of every lane with `ok` to components base .. base+14 of its gaussian's
// gradient row - one 64-B atomic request per record, nothing returned, so the requests drain behind the wave's arithmetic
// (a lane adding its own 15 values issues 15 x 64 scattered 4-B requests per instruction row; the earlier record paths -
// per-block buckets + counting-sort reduce, per-wave record logs + an apply kernel - are measured in DESIGN.md 4 and gone).
// Wave-uniform call; `stage` = 64 x 4 float4.
EGR_DI void wide_add_wave(const DeviceView &v, bool ok, uint32_t pos, const float (&r)[15], uint32_t base, float4 *stage) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    if (ok) {
        stage[4 * lane + 0] = make_float4(u2f(pos), r[0], r[1], r[2]);
        stage[4 * lane + 1] = make_float4(r[3], r[4], r[5], r[6]);
        stage[4 * lane + 2] = make_float4(r[7], r[8], r[9], r[10]);
        stage[4 * lane + 3] = make_float4(r[11], r[12], r[13], r[14]);
    }
    const unsigned long long okm = __ballot(ok);
    EGR_BWD_SYNC();
    const float *sf = reinterpret_cast<const float *>(stage);
#pragma unroll 4
    for (int pass = 0; pass < 16; pass++) {
        const int src = 4 * pass + (lane >> 4), c = lane & 15;
        if (((okm >> (4 * pass)) & 0xFull) == 0ull) continue; // wave-uniform: none of these four records exists
        const float x = sf[16 * src + c];
        const uint32_t p = f2u(sf[16 * src]);
        if (((okm >> src) & 1ull) && c != 0 && x != 0.0f) atomicAdd(v.grad_rows + (size_t)p * EGR_ROW_STRIDE + base + (uint32_t)(c - 1), x);
    }
    EGR_BWD_SYNC();
}

// Flush of the primary step's LDS table: every used slot leaves as two 16-lane records (components 0-14 and 15-21 of the row).
EGR_DI uint32_t grad_table_flush(const DeviceView &v, uint32_t *gt_keys, float *gt_vals, float4 *stage, int lane) {
    uint32_t sent = 0u; // records (two per used slot)
    EGR_BWD_SYNC();
    for (int s0 = 0; s0 < EGR_GT_SLOTS; s0 += EGR_WAVE) { // wave-uniform: wide_add_wave is a wave-level operation
        const int s = min(s0 + lane, EGR_GT_SLOTS - 1); // (a slot count that is no multiple of 64: the lanes beyond the table idle on its last slot)
        const bool in_table = s0 + lane < EGR_GT_SLOTS;
        const uint32_t pos = in_table ? gt_keys[s] : EGR_GT_EMPTY;
        const bool valid = pos != EGR_GT_EMPTY;
        if (in_table) gt_keys[s] = EGR_GT_EMPTY;
        float lo[15], hi[15];
#pragma unroll
        for (int c = 0; c < 15; c++) {
            lo[c] = valid ? gt_vals[s * EGR_GT_STRIDE + c] : 0.0f;
            if (valid) gt_vals[s * EGR_GT_STRIDE + c] = 0.0f; // (an unused slot holds zeros already)
        }
#pragma unroll
        for (int c = 0; c < 15; c++) {
            hi[c] = 0.0f;
            if (15 + c < EGR_GT_COMPS) {
                hi[c] = valid ? gt_vals[s * EGR_GT_STRIDE + 15 + c] : 0.0f;
                if (valid) gt_vals[s * EGR_GT_STRIDE + 15 + c] = 0.0f;
            }
        }
        if (__ballot(valid) == 0ull) continue;
        wide_add_wave(v, valid, pos, lo, 0u, stage);
        wide_add_wave(v, valid, pos, hi, 15u, stage);
        sent += 2u * (uint32_t)__popcll(__ballot(valid));
    }
    EGR_BWD_SYNC();
    return sent;
}

// The 22 gradient components of the PRIMARY hits of one hit row (one per lane; `pos` = record index) on their way into the wave's LDS table, in two
// calls: primary_presum (1), then primary_table_add (2, 3), which returns true for a lane whose contribution found no slot and has to leave
// as two records of its own.
//  (1) Primary tiles are coherent: the pixel to the right / below very often composites the SAME gaussian at the same hit index, so equal
//      neighbours are summed in registers first (two DPP levels: x neighbour, then y neighbour) and only the surviving lane touches the table.
//  (2) The table, keyed by record index. LDS float atomics retire about ONE LANE PER CLOCK PER CU, shared by all resident waves: 22
//      ds_add_f32 per hit row (1408 lane-operations) made the table half of the backward chain (round 3, per-phase stamps: 12k of 23k
//      wave-cycles per row on the dense-init cloud). The table is private to this wave, so only lanes of the SAME row can collide, and only
//      on the same gaussian: a lane finds its slot with plain reads (one CAS when it has to create it), the lanes of a slot elect one of them
//      per round through a claim word, and the winner adds its 22 components with plain ds_read / ds_write (conflict-free: odd slot stride)
//      - the others follow in the next round (after the register-level pre-sums few slots see more than one lane).
//  (3) A hit that finds no slot within 8 probes: the table is (as good as) full. A tile of the dense-init cloud composites more gaussians
//      (~ 100-130) than the table holds, and every hit of a gaussian without a slot used to leave the wave as two 64-B records of its own -
//      a resident gaussian costs two per TILE. The hits of a tile arrive far to near, layer by layer, so what a full table holds is mostly
//      done with: the wave empties it (two records per used slot, counted in `records`) and looks its hits up again.
// (1): called by the lanes that hold a hit (inside their divergent block: `__ballot(true)` is the set of them); returns whether this lane still owns a contribution.
EGR_DI bool primary_presum(const uint32_t pos, float (&gx)[EGR_GT_COMPS], const int lane) {
    bool mine = true;
    {
        const unsigned long long em = __ballot(true);
#define EGR_FETCH_DPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (int)(x), ctrl, 0xF, 0xF, false)
#define EGR_FETCH_SWZ(x, pat) __builtin_amdgcn_ds_swizzle((int)(x), pat) // bit-mask mode: lane ^ xor inside a group of 32 (no LDS memory touched)
#define EGR_FETCH_X32(x, unused) __shfl_xor((int)(x), 32)
#define EGR_FETCH_DPP4(x, unused) __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, (int)(x), 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false)
#define EGR_COMBINE(d, FETCH, arg)                                                                            \
    {                                                                                                          \
        const bool pact = ((em >> ((uint32_t)lane ^ (d))) & 1ull) != 0ull;                                     \
        const uint32_t ppos = (uint32_t)FETCH(pos, arg), pmine = (uint32_t)FETCH(mine ? 1 : 0, arg);           \
        const bool same = pact && pmine != 0u && mine && ppos == pos;                                          \
        const bool take = same && ((uint32_t)lane & (d)) == 0u;                                                \
        _Pragma("unroll") for (int c = 0; c < EGR_GT_COMPS; c++) {                                             \
            const float o = __int_as_float(FETCH(__float_as_int(gx[c]), arg));                                 \
            gx[c] += take ? o : 0.0f;                                                                          \
        }                                                                                                      \
        if (same && !take) mine = false;                                                                       \
    }
        EGR_COMBINE(1u, EGR_FETCH_DPP, 0xB1)  // quad_perm [1,0,3,2]: lane ^ 1
        EGR_COMBINE(8u, EGR_FETCH_DPP, 0x128) // row_ror 8: lane ^ 8
#undef EGR_COMBINE
#undef EGR_FETCH_DPP
#undef EGR_FETCH_SWZ
#undef EGR_FETCH_X32
#undef EGR_FETCH_DPP4
    }
    return mine;
}
// (2), (3): wave-level; `want` = this lane holds a contribution (primary_presum) that looks for a table slot.
EGR_DI bool primary_table_add(const DeviceView &v, const bool want, const uint32_t pos, const float (&gx)[EGR_GT_COMPS], uint32_t *gt_keys, float *gt_vals, uint32_t *gt_claim, float4 *stage,
                              const int lane, uint32_t &records) {
    bool found = false;
    uint32_t slot = 0u;
    auto probe = [&]() {
        slot = ((((pos * 2654435761u) >> 16) * (uint32_t)EGR_GT_SLOTS) >> 16); // keyed by record index (multiply-shift: any slot count)
        found = false;
#pragma unroll 1
        for (int pr = 0; pr < (want ? 8 : 0); pr++) {
            uint32_t key = *reinterpret_cast<volatile uint32_t *>(&gt_keys[slot]);
            if (key == EGR_GT_EMPTY) key = atomicCAS(&gt_keys[slot], EGR_GT_EMPTY, pos), key = key == EGR_GT_EMPTY ? pos : key; // (another lane of this row may have taken it meanwhile)
            if (key == pos) { found = true; break; }
            slot = slot + 1u == (uint32_t)EGR_GT_SLOTS ? 0u : slot + 1u;
        }
    };
    probe();
    if (__ballot(want && !found) != 0ull) {
        records += grad_table_flush(v, gt_keys, gt_vals, stage, lane);
        probe();
    }
    bool pending = found; // this lane's contribution still has to be added to its table slot
    while (__ballot(pending) != 0ull) { // the rounds of the table update (wave-uniform loop)
        if (pending) gt_claim[slot] = (uint32_t)lane;
        EGR_BWD_SYNC();
        const bool win = pending && gt_claim[slot] == (uint32_t)lane;
        if (win) {
            float *cell = gt_vals + slot * EGR_GT_STRIDE;
            float cur[EGR_GT_COMPS];
#pragma unroll
            for (int c = 0; c < EGR_GT_COMPS; c++) cur[c] = cell[c];
#pragma unroll
            for (int c = 0; c < EGR_GT_COMPS; c++) cell[c] = cur[c] + gx[c];
            pending = false;
        }
        EGR_BWD_SYNC();
    }
    return want && !found; // (still no slot: the 22 components leave as two 16-lane records, like a flushed slot)
}

// ---- the geometry half of B2 for one hit (backward_pass.cu:151-205): components 0 .. 10 of its gradient (opacity, scale, mean,
// rotation) from the record position, the ray, the live (.., opacity, sigma) quarter and dL/dalpha. Independent of the other hits of
// the ray - the sequential half (suffix sums -> dL/dalpha) is the caller's.
template <class GX> EGR_DI void hit_geometry_fn(const DeviceView &v, const float exp_power, const float eps_scale_grad, uint32_t pos, const float4 &a2, const f3 &ro, const f3 &rd, float dL_dalpha, GX &gx) {
    const float opacity = a2.z, scaling_factor = a2.w;
    // recompute the local hit exactly as the forward did
    const float4 W0 = v.inst_w[4 * pos], W1 = v.inst_w[4 * pos + 1], W2 = v.inst_w[4 * pos + 2];
    const float4 M0 = v.inst_m[4 * pos], M1 = v.inst_m[4 * pos + 1], M2 = v.inst_m[4 * pos + 2], qu = v.inst_m[4 * pos + 3]; // (requested with W: left where they are used, the compiler issues them ~ 500 instructions later and the row waits a second time)
    f3 lo, ld, dhat, u;
    float t_unused;
    bool behind_unused, outside_unused;
    candidate_geometry(W0, W1, W2, ro, rd, lo, ld, dhat, t_unused, u, behind_unused, outside_unused);
    const f3 local_hit = u * scaling_factor;
    const float sq_norm = dot(local_hit, local_hit);
    const float gaussval = egr_gaussian_sq(sq_norm, exp_power);

    float d_opacity = EGR_MAX_ALPHA * dL_dalpha * gaussval; // :151-152
    d_opacity = d_opacity * opacity * (1.0f - opacity);
    const float dL_dgaussval = EGR_MAX_ALPHA * dL_dalpha * opacity; // :155-158
    const float dL_dsq_norm = gaussval * pow_exp_m1(sq_norm, exp_power);
    const f3 dL_dx_local = (-local_hit * dL_dsq_norm) * dL_dgaussval;
    const f3 dL_dx_world = mk3(dot(mk3(W0.x, W1.x, W2.x), dL_dx_local), dot(mk3(W0.y, W1.y, W2.y), dL_dx_local),
                               dot(mk3(W0.z, W1.z, W2.z), dL_dx_local)) * scaling_factor; // :161-167
    const f3 dl2w0 = -dL_dx_world.x * local_hit, dl2w1 = -dL_dx_world.y * local_hit, dl2w2 = -dL_dx_world.z * local_hit;
    const f3 d_mean = -dL_dx_world;
    const f3 scaling = mk3(M0.w, M1.w, M2.w); // exp(scale), stored by k_instances
    const f3 den = mk3(scaling.x * scaling_factor + eps_scale_grad, scaling.y * scaling_factor + eps_scale_grad,
                       scaling.z * scaling_factor + eps_scale_grad);
    // (nine quotients over three denominators, further down four over |q| and two reciprocals: each denominator's reciprocal is refined once and every
    // quotient corrected from it - egr_div_rn, the results of `/` - instead of seventeen full expansions of a division)
    const f3 rden = mk3(egr_rcp_refined(den.x), egr_rcp_refined(den.y), egr_rcp_refined(den.z));
    const f3 rot_0 = egr_div3_rn(mk3(M0.x, M0.y, M0.z), den, rden), rot_1 = egr_div3_rn(mk3(M1.x, M1.y, M1.z), den, rden), rot_2 = egr_div3_rn(mk3(M2.x, M2.y, M2.z), den, rden); // :178-180
    const f3 d_scale = (dl2w0 * rot_0 + dl2w1 * rot_1 + dl2w2 * rot_2) * scaling; // :181-182
    const f3 dr0 = dl2w0 * scaling, dr1 = dl2w1 * scaling, dr2 = dl2w2 * scaling; // :185-187
    const float qn = egr_sqrt_rn(qu.x * qu.x + qu.y * qu.y + qu.z * qu.z + qu.w * qu.w);
    const float rqn = egr_rcp_refined(qn);
    const float r = egr_div_rn(qu.x, qn, rqn), x = egr_div_rn(qu.y, qn, rqn), y = egr_div_rn(qu.z, qn, rqn), z = egr_div_rn(qu.w, qn, rqn); // activations.cu:66-69
    const float dL_dr = 2.f * x * (dr2.y - dr1.z) + 2.f * y * (dr0.z - dr2.x) + 2.f * z * (dr1.x - dr0.y); // :194-205
    const float dL_dx = -4.f * x * (dr1.y + dr2.z) + 2.f * y * (dr0.y + dr1.x) + 2.f * z * (dr0.z + dr2.x) + 2.f * r * (dr2.y - dr1.z);
    const float dL_dy = 2.f * x * (dr0.y + dr1.x) - 4.f * y * (dr0.x + dr2.z) + 2.f * z * (dr1.z + dr2.y) + 2.f * r * (dr0.z - dr2.x);
    const float dL_dz = 2.f * x * (dr0.z + dr2.x) + 2.f * y * (dr1.z + dr2.y) - 4.f * z * (dr0.x + dr1.y) + 2.f * r * (dr1.x - dr0.y);
    const float dd = dL_dr * qu.x + dL_dx * qu.y + dL_dy * qu.z + dL_dz * qu.w; // activations.cu:71-73
    const float qn3 = qn * qn * qn;
    const float inv3 = egr_div_rn(1.0f, qn3, egr_rcp_refined(qn3)), inv1 = egr_div_rn(1.0f, qn, rqn);

    // :210-220 flush: into the LDS table when a slot is found within 8 probes, else straight to global
    const float d_rot0 = dd * -qu.x * inv3 + dL_dr * inv1, d_rot1 = dd * -qu.y * inv3 + dL_dx * inv1;
    const float d_rot2 = dd * -qu.z * inv3 + dL_dy * inv1, d_rot3 = dd * -qu.w * inv3 + dL_dz * inv1;
    gx[GC_OPA] = d_opacity, gx[GC_SCALE] = d_scale.x, gx[GC_SCALE + 1] = d_scale.y, gx[GC_SCALE + 2] = d_scale.z;
    gx[GC_MEAN] = d_mean.x, gx[GC_MEAN + 1] = d_mean.y, gx[GC_MEAN + 2] = d_mean.z;
    gx[GC_ROT] = d_rot0, gx[GC_ROT + 1] = d_rot1, gx[GC_ROT + 2] = d_rot2, gx[GC_ROT + 3] = d_rot3;
}

// One batch of 64 hits of a bounce step's queue (backward_task.inc: pass 2), lane = hit: the geometry gradient from the hit's record index, the
// ray and dL/dalpha, the radiance part from the ray's dL/drgb, all 15 components out as one record. Queue, rays and radiance gradients are the
// LDS of the wave that OWNS the tile; `stage` is the executing wave's. Returns the records sent.
EGR_DI uint32_t bounce_batch(const DeviceView &v, const float exp_power, const float eps_scale_grad, const uint4 *bitems, const float *bdl, const float *bray, uint32_t i0, uint32_t nitems,
                             float4 *stage, int lane) {
    const bool valid = i0 + (uint32_t)lane < nitems;
    uint32_t dpos = 0;
    float gx[15];
#pragma unroll
    for (int c = 0; c < 15; c++) gx[c] = 0.0f;
    if (valid) {
        // (everything a hit needs beyond its gaussian's records travels in the queue or sits in LDS: the fetches of this pass - live quarter,
        // W, M - are ONE round trip; it used to re-read the arena row first and the ray from the state)
        const uint4 item = bitems[i0 + (uint32_t)lane];
        const uint32_t ray = item.x, pos = item.z;
        const float weight = u2f(item.w);
        const float4 a2 = v.inst_w[4 * pos + 3];
        const f3 iro = mk3(bray[ray], bray[EGR_WAVE + ray], bray[2 * EGR_WAVE + ray]), ird = mk3(bray[3 * EGR_WAVE + ray], bray[4 * EGR_WAVE + ray], bray[5 * EGR_WAVE + ray]);
        hit_geometry_fn(v, exp_power, eps_scale_grad, pos, a2, iro, ird, u2f(item.y), gx);
        gx[GC_RGB] = bdl[ray] * weight, gx[GC_RGB + 1] = bdl[EGR_WAVE + ray] * weight, gx[GC_RGB + 2] = bdl[2 * EGR_WAVE + ray] * weight;
        gx[GC_WEIGHT] = weight;
        dpos = pos;
    }
    wide_add_wave(v, valid, dpos, gx, 0u, stage);
    return (uint32_t)__popcll(__ballot(valid));
}

// Teams of the backward chain (under-filled ranks of a partition): per wave a ticket (epoch << 20 | batches << 10 | next batch) for the chunk of
// bounce hits it has queued, the number of items, and the batches done. No list order is at stake here - gradients are atomic adds.
#ifndef EGR_BWD_TEAM
#define EGR_BWD_TEAM 4 // waves per workgroup of the backward chain's team build
#endif
template <int TEAM> struct BwdTeamShared {
    uint32_t done;
    uint32_t ticket[TEAM], nitems[TEAM], finished[TEAM];
};

// The backward half of the fused per-tile chain (see k_forward_chain): a wave takes a tile's backward through all its steps - the
// primary step (22 gradient components per hit through the LDS table), then the bounce steps, last bounce first (15 components per
// hit, straight out as wide adds; in the team build their batches can be taken by team mates). Per-step code: backward_task.inc.
#ifndef EGR_BWD_WAVES
#define EGR_BWD_WAVES 3
#endif
// Between the two chains of a grad launch: the order in which the backward chain takes this strand's tasks. The forward chain knows what a
// tile's backward will cost (its hit rows and hits), and a persistent-wave kernel ends with a tail as long as the tiles that START LATE and
// RUN LONG: per-task stamps of the whole image put the backward chain at 2807 us where a longest-first list schedule of the same task times
// needs 2289 us (dense-init; by this proxy 2323 us). One workgroup per queue chunk (wave_next_task: 8 chunks = compact image blocks, one
// per XCD) sorts ITS tasks by descending cost (stable counting sort over EGR_ORDER_BUCKETS = 16 cost classes) - a chunk's tiles stay on its XCD, only their order changes.
#ifndef EGR_ORDER_BUCKETS
#define EGR_ORDER_BUCKETS 16 // cost classes of the backward order
#endif
#ifndef EGR_ORDER_SHIFT
#define EGR_ORDER_SHIFT 7 // a class spans 2^7 = 128 units of cost (~ microseconds)
#endif
__global__ void __launch_bounds__(256) k_order_backward(DeviceView v) {
    // STABLE: inside a cost class the tiles keep the order of the chunk (a Z-curve over a compact image block: neighbours in time touch neighbouring records)
    __shared__ uint32_t cnt[EGR_ORDER_BUCKETS][256 + 1];
    const uint32_t chunk = ((v.task_count + 7u) / 8u + 3u) & ~3u, q = blockIdx.x;
    const uint32_t beg = q * chunk, end = min(beg + chunk, v.task_count);
    if (beg >= end) return;
    const uint32_t t = threadIdx.x, n = end - beg, per = (n + 255u) / 256u;
    const uint32_t i0 = min(t * per, n), i1 = min(i0 + per, n); // thread t owns the contiguous items [i0, i1) of the chunk
    auto bucket = [](uint32_t cost) { return (uint32_t)EGR_ORDER_BUCKETS - 1u - min(cost >> EGR_ORDER_SHIFT, (uint32_t)EGR_ORDER_BUCKETS - 1u); }; // (descending: costliest first)
    for (int b = 0; b < EGR_ORDER_BUCKETS; b++) cnt[b][t] = 0u;
    for (uint32_t i = i0; i < i1; i++) cnt[bucket(v.task_cost[v.task_begin + beg + i])][t]++;
    __syncthreads();
    if (t < (uint32_t)EGR_ORDER_BUCKETS) { // exclusive scan over the threads, one cost class per thread
        uint32_t acc = 0u;
        for (int j = 0; j < 256; j++) {
            const uint32_t c = cnt[t][j];
            cnt[t][j] = acc, acc += c;
        }
        cnt[t][256] = acc;
    }
    __syncthreads();
    uint32_t start[EGR_ORDER_BUCKETS]; // where this thread's items of every class go
    uint32_t acc = 0u;
#pragma unroll
    for (int b = 0; b < EGR_ORDER_BUCKETS; b++) start[b] = acc + cnt[b][t], acc += cnt[b][256];
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t bk = bucket(v.task_cost[v.task_begin + beg + i]);
        uint32_t at = 0u;
#pragma unroll
        for (int b = 0; b < EGR_ORDER_BUCKETS; b++)
            if ((uint32_t)b == bk) at = start[b]++;
        v.bwd_order[v.task_begin + beg + at] = v.task_begin + beg + i;
    }
}

// A team mate without tiles: take batches of the chunks its team mates have open (backward_task.inc: the ticket) until all waves are through.
template <int TEAM> EGR_DI uint32_t bwd_team_help(const DeviceView &v, const float exp_power, const float eps_scale_grad, BwdTeamShared<TEAM> &bteam, float (*gt_vals_all)[EGR_GT_STRIDE * EGR_GT_SLOTS],
                                                  float4 *stage, const int wv, const int lane) {
    uint32_t records = 0u;
    while (uniform_u32(lds_peek(&bteam.done)) < (uint32_t)TEAM) {
        bool any = false;
        for (int dlt = 1; dlt < TEAM; dlt++) {
            const int w = (wv + dlt) % TEAM;
            const uint32_t peek = uniform_u32(lds_peek(&bteam.ticket[w]));
            if ((peek & 0x3FFu) >= ((peek >> 10) & 0x3FFu)) continue; // nothing open there
            uint32_t tk = 0u;
            if (lane == 0) tk = atomicAdd(&bteam.ticket[w], 1u); // (epoch, batches and the batch taken come back in ONE word)
            tk = uniform_u32(tk);
            const uint32_t bi = tk & 0x3FFu, nbatch = (tk >> 10) & 0x3FFu;
            if (bi >= nbatch) continue;
            // the owner does not touch its queue, rays or radiance gradients before `finished` says that every batch of the chunk is done
            const uint32_t nitems = uniform_u32(lds_peek(&bteam.nitems[w]));
            const uint4 *bitems = reinterpret_cast<const uint4 *>(gt_vals_all[w]);
            records += bounce_batch(v, exp_power, eps_scale_grad, bitems, gt_vals_all[w] + 16 * EGR_WAVE, gt_vals_all[w] + 19 * EGR_WAVE, bi * (uint32_t)EGR_WAVE, nitems, stage, lane);
            if (lane == 0) atomicAdd(&bteam.finished[w], 1u);
            any = true;
        }
        if (!any) __builtin_amdgcn_s_sleep(2);
    }
    return records;
}

template <int TEAM> __global__ void __launch_bounds__(EGR_WAVE * TEAM) __attribute__((amdgpu_waves_per_eu(EGR_BWD_WAVES, EGR_BWD_WAVES))) k_backward_chain(DeviceView v) {
    const int lane = threadIdx.x & (EGR_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ uint32_t gt_keys_all[TEAM][EGR_GT_SLOTS];
    __shared__ __attribute__((aligned(16))) float gt_vals_all[TEAM][EGR_GT_STRIDE * EGR_GT_SLOTS];
    __shared__ uint32_t gt_claim_all[TEAM][EGR_GT_SLOTS]; // which lane adds to a slot in this round (backward_task.inc)
    __shared__ float4 stage_all[TEAM][4 * EGR_WAVE];     // records on their way out (wide_add_wave)
    __shared__ BwdTeamShared<TEAM> bteam;
    uint32_t *const gt_keys = gt_keys_all[wv], *const gt_claim = gt_claim_all[wv];
    float *const gt_vals = gt_vals_all[wv];
    float4 *const stage = stage_all[wv];
    if (threadIdx.x == 0) bteam.done = 0u;
    if (threadIdx.x < TEAM) bteam.ticket[threadIdx.x] = 0u, bteam.nitems[threadIdx.x] = 0u, bteam.finished[threadIdx.x] = 0u;
    uint32_t bepoch = 0u;
    // the two queues of the bounce steps live in the table's memory: the table is empty (flushed, all zero) while a tile's bounce steps run -
    // they come before its primary step - and the words they dirtied are cleared again before that step (backward_task.inc).
#define EGR_BQ_FLOATS (25 * EGR_WAVE) // floats of the table's memory the bounce steps use
    static_assert(EGR_GT_STRIDE * EGR_GT_SLOTS >= EGR_BQ_FLOATS, "the bounce queues must fit into the table");
    uint4 *bitems = reinterpret_cast<uint4 *>(gt_vals);  // [4 x 64] bounce steps: (ray, dL/dalpha, record, weight) of the hits of a chunk of four rows
    float *bdl = gt_vals + 16 * EGR_WAVE;                // [3 x 64] bounce steps: the rays' radiance gradient
    float *bray = gt_vals + 19 * EGR_WAVE;               // [6 x 64] bounce steps: the rays (origin, direction)
    for (int s = lane; s < EGR_GT_SLOTS; s += EGR_WAVE) gt_keys[s] = EGR_GT_EMPTY;
    for (int s = lane; s < EGR_GT_STRIDE * EGR_GT_SLOTS; s += EGR_WAVE) gt_vals[s] = 0.0f;
    __syncthreads(); // the kernel's only workgroup barrier (a team's waves run independently from here on)
    const float exp_power = *v.cfg.exp_power;
    const float eps_scale_grad = *v.cfg.eps_scale_grad;
    const int num_bounces = min(*v.cfg.num_bounces, EGR_MAX_BOUNCES);
    const float view_size = tanf(*v.cam.vertical_fov_radians / 2.0f); // (primary_direction)
    uint32_t cur_q = blockIdx.x & 7u;
    uint32_t records = 0u; // 64-B gradient records this wave sent: bounce hits, primary hits without a table slot (two each), flushed table slots (two each) (egr_counters::bucket_records)

    for (;;) {
        const uint32_t tq = wave_next_task(v.queues + 8 * EGR_QUEUE_STRIDE, v.task_count, cur_q, lane);
        if (tq == 0xFFFFFFFFu) break;
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 8 // diagnostic build: stamps of a task's BACKWARD chain in its first pixels (tools/bwd_times.py)
        const unsigned long long bw_t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long bw_t1 = 0ull;
        uint32_t bw_rows0 = 0u;
#endif
        // The steps of a tile are independent in the backward (each reads its own arena chain and the forward's state, all gradients are
        // atomic adds), so their order is free: the PRIMARY step goes first and the bounce steps last - the bounce steps' batches are what
        // team mates can take (backward_task.inc), and team mates only have time once their own tiles are through, i.e. late in a heavy tile.
        bool table_dirty = false; // (wave-uniform) a bounce step used the table's memory for its queues
        {
            constexpr bool PRIMARY = true;
            const int step = 0;
            do {
#include "backward_task.inc"
            } while (false);
        }
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 8
        bw_t1 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int step = num_bounces; step >= 1; step--) {
            constexpr bool PRIMARY = false;
            do {
#include "backward_task.inc"
            } while (false);
        }
        if (table_dirty) { // (the table's memory held the bounce steps' queues: empty again for the next tile's primary step)
            EGR_BWD_SYNC();
            for (int s = lane; s < EGR_BQ_FLOATS; s += EGR_WAVE) gt_vals[s] = 0.0f;
            EGR_BWD_SYNC();
        }
#if defined(EGR_TASK_TIMES) && EGR_TASK_TIMES == 8
        {
            const unsigned long long bw_t2 = __builtin_amdgcn_s_memrealtime();
            const TaskGeom btg = task_geom(v, v.bwd_order ? v.bwd_order[v.task_begin + tq] : v.task_begin + tq, lane);
            if (btg.inside) {
                if (lane == 0) v.stats.num_traversed_per_pixel[btg.pixel_id] = (int32_t)(bw_t0 & 0x7FFFFFFFull), v.stats.num_accumulated_per_pixel[btg.pixel_id] = (int32_t)(bw_t2 & 0x7FFFFFFFull);
                if (lane == 1) v.stats.num_traversed_per_pixel[btg.pixel_id] = (int32_t)(bw_t1 & 0x7FFFFFFFull), v.stats.num_accumulated_per_pixel[btg.pixel_id] = (int32_t)bw_rows0;
            }
        }
#endif
    }
    if constexpr (TEAM > 1) {
        // no tiles left: this wave takes batches of bounce hits its team mates have queued until all of them are through
        if (lane == 0) atomicAdd(&bteam.done, 1u);
        records += bwd_team_help<TEAM>(v, exp_power, eps_scale_grad, bteam, gt_vals_all, stage, wv, lane);
    }
    if (lane == 0 && records) atomicAdd(v.control + CW_BUCKET_RECORDS, records);
}

// Last backward kernel: one thread per gaussian id adds its gradient row (a 128-B line at its record position) to the
// caller's gradient tensors - coalesced on the tensor side - and empties the row for the next launch.
// OVERWRITE (egr_set_grad_overwrite: the tensors are a per-launch buffer that is all-reduced over the ranks): the launch's sums are
// STORED, rows without a contribution store zeros - the caller neither clears the buffer nor pays the read half of a "+=".
template <bool OVERWRITE> __global__ void __launch_bounds__(256) k_grad_gather(DeviceView v) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= v.n) return;
    float4 *row = reinterpret_cast<float4 *>(v.grad_rows + (size_t)v.pos_of_gid[gid] * EGR_ROW_STRIDE);
    float x[24];
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const float4 r = row[q];
        x[4 * q] = r.x, x[4 * q + 1] = r.y, x[4 * q + 2] = r.z, x[4 * q + 3] = r.w;
        any |= (f2u(r.x) | f2u(r.y) | f2u(r.z) | f2u(r.w)) << 1; // ignore the sign bit: -0 is empty too
    }
    if (!OVERWRITE && any == 0) return;
    if (any != 0) {
#pragma unroll
        for (int q = 0; q < 6; q++) row[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const egr_gaussians &g = v.g;
    auto put = [](float *p, float val) {
        if (OVERWRITE) *p = val;
        else *p += val;
    };
    put(g.dL_dopacity + gid, x[GC_OPA]);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        put(g.dL_dscale + 3 * gid + a, x[GC_SCALE + a]), put(g.dL_dmean + 3 * gid + a, x[GC_MEAN + a]), put(g.dL_drgb + 3 * gid + a, x[GC_RGB + a]);
        put(g.dL_dnormal + 3 * gid + a, x[GC_NORMAL + a]), put(g.dL_df0 + 3 * gid + a, x[GC_F0 + a]);
    }
#pragma unroll
    for (int a = 0; a < 4; a++) put(g.dL_drotation + 4 * gid + a, x[GC_ROT + a]);
    put(g.dL_droughness + gid, x[GC_ROUGH]);
    put(g.total_weight + gid, x[GC_WEIGHT]);
}

// ---------------------------------------------------------------------------------------------------------
// R6: output write / accumulation (shaders.cu:163-169, framebuffer.h:104-143); no-grad launches only
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(EGR_WAVE) k_finish(DeviceView v) {
    const int lane = threadIdx.x;
    const bool accumulate = *v.cfg.accumulate_samples != 0;
    const float cnt = accumulate ? (float)(*v.fb.accumulated_sample_count + 1) : 1.0f;
    const size_t P = v.num_pixels;
    for (uint32_t task = v.task_begin + blockIdx.x; task < v.task_begin + v.task_count; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        if (!tg.inside) continue;
        const StateRef S = state_of(v, task, lane);
        const uint32_t steps = f2u(S.ld(F_STEPS));
        f3 final = mk3(0, 0, 0);
        for (int s = 0; s < EGR_NSTEPS; s++) {
            const bool done = (uint32_t)s < steps;
            f3 rgb = done ? S.ld3(SF(s, S_RGB)) : mk3(0, 0, 0), n = done ? S.ld3(SF(s, S_NORMAL)) : mk3(0, 0, 0);
            f3 f0 = done ? S.ld3(SF(s, S_F0)) : mk3(0, 0, 0);
            float depth = done ? S.ld(SF(s, S_DEPTH)) : 0.0f, rough = done ? S.ld(SF(s, S_ROUGH)) : 0.0f;
            float T = done ? S.ld(SF(s, S_T)) : 1.0f, Ttot = done ? S.ld(SF(s, S_TTOT)) : 1.0f;
            f3 no = done ? S.ld3(SF(s, S_NEXT_O)) : mk3(0, 0, 0), nd = done ? S.ld3(SF(s, S_NEXT_D)) : mk3(0, 0, 0);
            const size_t q = tg.pixel_id + P * (size_t)s;
            if (accumulate) { // framebuffer.h:104-128
#define EGR_ACC3(buf, val) { float *a_ = v.fb.buf + 3 * q; a_[0] += val.x, a_[1] += val.y, a_[2] += val.z; val = div_s(mk3(a_[0], a_[1], a_[2]), cnt); }
#define EGR_ACC1(buf, val) { float *a_ = v.fb.buf + q; a_[0] += val; val = a_[0] / cnt; }
                EGR_ACC3(accumulated_rgb, rgb) EGR_ACC1(accumulated_transmittance, T) EGR_ACC1(accumulated_total_transmittance, Ttot)
                EGR_ACC1(accumulated_depth, depth) EGR_ACC3(accumulated_normal, n) EGR_ACC3(accumulated_f0, f0) EGR_ACC1(accumulated_roughness, rough)
#undef EGR_ACC3
#undef EGR_ACC1
            }
            final = final + rgb;
            float *o;
            o = v.fb.output_rgb + 3 * q, o[0] = rgb.x, o[1] = rgb.y, o[2] = rgb.z;
            o = v.fb.output_normal + 3 * q, o[0] = n.x, o[1] = n.y, o[2] = n.z;
            o = v.fb.output_f0 + 3 * q, o[0] = f0.x, o[1] = f0.y, o[2] = f0.z;
            o = v.fb.output_ray_origin + 3 * q, o[0] = no.x, o[1] = no.y, o[2] = no.z;
            o = v.fb.output_ray_direction + 3 * q, o[0] = nd.x, o[1] = nd.y, o[2] = nd.z;
            v.fb.output_depth[q] = depth, v.fb.output_roughness[q] = rough;
            v.fb.output_transmittance[q] = T, v.fb.output_total_transmittance[q] = Ttot;
        }
        float *o = v.fb.output_final + 3 * (size_t)tg.pixel_id;
        o[0] = final.x, o[1] = final.y, o[2] = final.z;
    }
}

// Diagnostic export (egr_debug_get_step_hits): composited hits per pixel and bounce step of the last GRAD launch, from the ray state
// the backward chain reads (S_NHITS); steps a ray did not execute report 0. Pixels outside this rank's partition are not written.
__global__ void __launch_bounds__(EGR_WAVE) k_export_step_hits(DeviceView v, int32_t *__restrict__ out) {
    const int lane = threadIdx.x;
    for (uint32_t task = blockIdx.x; task < v.num_tasks; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        if (!tg.inside) continue;
        const StateRef S = state_of(v, task, lane);
        const uint32_t steps = f2u(S.ld(F_STEPS));
        for (int s = 0; s < EGR_NSTEPS; s++) out[(size_t)s * v.num_pixels + tg.pixel_id] = (uint32_t)s < steps ? (int32_t)f2u(S.ld(SF(s, S_NHITS))) : 0;
    }
}

// Diagnostic export (egr_debug_get_hit_sequence_hash): per pixel and step a hash of the ORDERED sequence of gaussian ids the last GRAD launch composited - the
// polynomial sum_i (id_i + 1) B^i mod 2^64 over the composite index i, evaluated back to front (Horner) along the step's arena chain, which is the order the
// backward walks; the CPU oracle evaluates the same sum front to back (oracle/egr_oracle.cpp: Outputs::hit_sequence_hash). Equal hashes = the same hits in the same order.
__global__ void __launch_bounds__(EGR_WAVE) k_export_hit_hash(DeviceView v, unsigned long long *__restrict__ out) {
    const int lane = threadIdx.x;
    constexpr unsigned long long B = 0x9E3779B97F4A7C15ull;
    for (uint32_t task = blockIdx.x; task < v.num_tasks; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        const StateRef S = state_of(v, task, lane);
        const uint32_t steps = tg.inside ? f2u(S.ld(F_STEPS)) : 0u;
        for (int s = 0; s < EGR_NSTEPS; s++) {
            uint32_t blk = v.task_last_block[(size_t)s * v.num_tasks + task];
            const uint32_t nhits = (tg.inside && (uint32_t)s < steps && blk != 0xFFFFFFFFu) ? f2u(S.ld(SF(s, S_NHITS))) : 0u;
            const uint32_t max_hits = wave_max_u32(nhits);
            unsigned long long h = 0ull;
            const uint32_t nblocks = (max_hits + EGR_HIT_BLOCK_ROWS - 1) / EGR_HIT_BLOCK_ROWS;
            for (uint32_t b = nblocks; b-- > 0 && blk != 0xFFFFFFFFu;) {
                const float4 *rows = v.hit_arena + (size_t)blk * (EGR_HIT_BLOCK_ROWS + 1) * EGR_WAVE;
                for (int row = EGR_HIT_BLOCK_ROWS - 1; row >= 0; row--)
                    if (b * EGR_HIT_BLOCK_ROWS + (uint32_t)row < nhits) h = h * B + (unsigned long long)v.gid_of_pos[f2u(rows[(size_t)(1 + row) * EGR_WAVE + lane].x)] + 1ull;
                blk = f2u(rows[0].x); // the block before this one (header row, lane 0's word)
            }
            if (tg.inside) out[(size_t)s * v.num_pixels + tg.pixel_id] = h;
        }
    }
}

// Camera upload: what gaussian_raytracer.py:94-100 does with ten tiny tensor kernels (R_blender = -R with column 0 negated back, three fill_ calls,
// set_pose's three copies) as ONE: R = the dataset's camera-to-world rotation (row-major 3x3), centre = camera_center, both device pointers.
__global__ void k_set_camera(egr_camera cam, const float *__restrict__ R, const float *__restrict__ centre, float fov, float znear, float zfar) {
    const int t = threadIdx.x;
    if (t < 9) {
        const int r = t / 3, c = t % 3;
        const float rb = c == 0 ? R[t] : -R[t]; // (-R)[:, 0] *= -1
        const_cast<float *>(cam.rotation_c2w)[t] = rb;
        const_cast<float *>(cam.rotation_w2c)[3 * c + r] = rb; // transpose
    }
    if (t < 3) const_cast<float *>(cam.origin)[t] = centre[t];
    if (t == 0) *const_cast<float *>(cam.vertical_fov_radians) = fov, *const_cast<float *>(cam.znear) = znear, *const_cast<float *>(cam.zfar) = zfar;
}

// Target upload (the caller's six `buf.copy_(val.moveaxis(0, -1))` of gaussian_raytracer.py:109-137 as ONE launch): CHW images -> the framebuffer's HWC
// target buffers, for the pixels of THIS rank's tiles only (a rank of an 8-way partition reads and writes an eighth of the 116 MB a whole
// 1080p frame moves); a missing image writes zeros, like the reference's `buf.zero_()`.
struct TargetImages {
    const float *chw[6]; // diffuse, specular, depth, normal, roughness, f0
};
__global__ void __launch_bounds__(EGR_WAVE) k_upload_targets(DeviceView v, TargetImages t) {
    const int lane = threadIdx.x;
    const size_t P = v.num_pixels;
    float *const dst[6] = {const_cast<float *>(v.fb.target_diffuse), const_cast<float *>(v.fb.target_specular), const_cast<float *>(v.fb.target_depth), const_cast<float *>(v.fb.target_normal), const_cast<float *>(v.fb.target_roughness), const_cast<float *>(v.fb.target_f0)}; // (the mirror of core/framebuffer.h declares the targets read-only for the launch)
    constexpr int ch[6] = {3, 3, 1, 3, 1, 3};
    for (uint32_t task = blockIdx.x; task < v.num_tasks; task += gridDim.x) {
        const TaskGeom tg = task_geom(v, task, lane);
        if (!tg.inside) continue;
#pragma unroll
        for (int b = 0; b < 6; b++)
#pragma unroll
            for (int c = 0; c < ch[b]; c++) dst[b][(size_t)ch[b] * tg.pixel_id + c] = t.chw[b] ? t.chw[b][(size_t)c * P + tg.pixel_id] : 0.0f;
    }
}

// The same for a tracer that owns the whole image (one rank): no tile geometry is needed, so the planes are read and the pixel-major
// buffers written 16 B per lane (four consecutive pixels per thread: 3 + 3 loads and stores for a colour buffer), 0.12 -> 0.05 ms per
// 1080p frame. Needs P % 4 == 0 and 16-B aligned planes (egr_upload_targets checks).
__global__ void __launch_bounds__(256) k_upload_targets_flat(DeviceView v, TargetImages t) {
    const size_t quads = v.num_pixels / 4, P = v.num_pixels;
    float *const dst[6] = {const_cast<float *>(v.fb.target_diffuse), const_cast<float *>(v.fb.target_specular), const_cast<float *>(v.fb.target_depth), const_cast<float *>(v.fb.target_normal), const_cast<float *>(v.fb.target_roughness), const_cast<float *>(v.fb.target_f0)};
    constexpr int ch[6] = {3, 3, 1, 3, 1, 3};
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int b = 0; b < 6; b++) {
            if (ch[b] == 1) {
                float4 a = t.chw[b] ? reinterpret_cast<const float4 *>(t.chw[b])[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4 *>(dst[b])[q] = a;
            } else {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a, c = a;
                if (t.chw[b]) {
                    a = reinterpret_cast<const float4 *>(t.chw[b])[q];
                    g = reinterpret_cast<const float4 *>(t.chw[b] + P)[q];
                    c = reinterpret_cast<const float4 *>(t.chw[b] + 2 * P)[q];
                }
                float4 *o = reinterpret_cast<float4 *>(dst[b]) + 3 * q;
                o[0] = make_float4(a.x, g.x, c.x, a.y);
                o[1] = make_float4(g.y, c.y, a.z, g.z);
                o[2] = make_float4(c.z, a.w, g.w, c.w);
            }
        }
    }
}

__global__ void k_copy3(const float *__restrict__ src, float *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

} // namespace

uint32_t egr_num_tasks_for_rank(const egr_context *c) {
    uint32_t mtx = (c->width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE, mty = (c->height + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    const uint32_t M = mtx * mty, world = (uint32_t)c->world, rank = (uint32_t)c->rank;
    // every block of `world` consecutive Z-order positions holds each rank once; in the last, partial block b the positions j < rem go
    // to ranks (j + b) % world (see egr_build_task_order)
    const uint32_t full = M / world, rem = M % world;
    const uint32_t extra = ((rank + world - full % world) % world) < rem ? 1u : 0u;
    return 4u * (full + extra); // in 8x8 tiles; x 2 / x 4 with smaller tasks (egr_make_view)
}

// Which rank owns which macro tile ("tile k -> GPU k mod world after a Morton shuffle"): the macro tiles are sorted along a
// Z-curve over (mx, my); the tile at position i of that order belongs to rank (i + i / world) % world. Every run of `world` consecutive
// positions - a compact 2-D block of the image (eight ranks: 4 x 2 macro tiles) - holds each rank exactly once, and the rotation by the
// block index keeps a rank from sitting at the same place of every block (two ranks: a checkerboard, not columns). The plain
// `tile index % world` of rounds 1-3 gave every rank vertical 16-pixel column stripes whenever the number of tile columns was a
// multiple of the world size (120 at 1920 px: 2, 4, 8).
// Mirrors: parallel.tile_owner (Python), Oracle::set_partition (the CPU checker, oracle/egr_oracle.cpp); exported to C hosts as egr_tile_owner.
// Order of this rank's macro tiles: sort by (XCD chunk block, Z-curve inside the block). The 8 chunks that
// wave_next_task hands to the 8 XCD queues are equal slices of this order, so each is a compact image block.
static std::vector<std::pair<uint32_t, uint32_t>> egr_macro_tile_zorder(uint32_t mtx, uint32_t mty) { // (Z-curve key, macro tile) of the whole image, sorted
    auto part1by1 = [](uint32_t x) {
        x &= 0xFFFFu;
        x = (x | (x << 8)) & 0x00FF00FFu, x = (x | (x << 4)) & 0x0F0F0F0Fu, x = (x | (x << 2)) & 0x33333333u, x = (x | (x << 1)) & 0x55555555u;
        return x;
    };
    std::vector<std::pair<uint32_t, uint32_t>> zorder;
    zorder.reserve((size_t)mtx * mty);
    for (uint32_t m = 0; m < mtx * mty; m++) zorder.push_back({part1by1(m % mtx) | (part1by1(m / mtx) << 1), m});
    std::sort(zorder.begin(), zorder.end());
    return zorder;
}
extern "C" int egr_tile_owner(int width, int height, int world, int tile_index) {
    if (width <= 0 || height <= 0 || world < 1) return -1;
    const uint32_t mtx = ((uint32_t)width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE, mty = ((uint32_t)height + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    if (tile_index < 0 || (uint32_t)tile_index >= mtx * mty) return -1;
    const auto zorder = egr_macro_tile_zorder(mtx, mty);
    for (size_t i = 0; i < zorder.size(); i++)
        if (zorder[i].second == (uint32_t)tile_index) return (int)((i + i / (size_t)world) % (size_t)world);
    return -1;
}
void egr_build_task_order(egr_context *c) {
    for (const auto &o : c->task_orders)
        if (o.rank == c->rank && o.world == c->world) { // built before: kernels in flight keep reading the table they were launched with
            c->task_macro = o.table;
            return;
        }
    if (c->task_orders.size() >= 8) { // a caller cycling through many partitions: drop the cache (nothing may still read the tables)
        EGR_HIP(hipDeviceSynchronize());
        for (auto &o : c->task_orders) egr_dev_free(c, o.table);
        c->task_orders.clear();
    }
    const uint32_t mtx = (c->width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE, mty = (c->height + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    const std::vector<std::pair<uint32_t, uint32_t>> zorder = egr_macro_tile_zorder(mtx, mty);
    std::vector<std::pair<uint64_t, uint32_t>> keyed;
    for (size_t b = 0; b * (size_t)c->world < zorder.size(); b++) { // this rank's share: one tile of every block of `world` Z-order positions
        const size_t i = b * (size_t)c->world + (size_t)(((uint32_t)c->rank + (uint32_t)c->world - (uint32_t)(b % (size_t)c->world)) % (uint32_t)c->world);
        if (i >= zorder.size()) continue;
        const uint32_t m = zorder[i].second;
        const uint32_t mx = m % mtx, my = m / mtx;
        const uint32_t bx = std::min(3u, mx * 4u / mtx), by = std::min(1u, my * 2u / mty); // 4 x 2 blocks of the image
        const uint64_t key = ((uint64_t)(by * 4u + bx) << 32) | zorder[i].first;
        keyed.push_back({key, m});
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> order(std::max<size_t>(keyed.size(), 1), 0u);
    for (size_t i = 0; i < keyed.size(); i++) order[i] = keyed[i].second;
    uint32_t *table = nullptr;
    egr_dev_alloc(c, table, order.size());
    EGR_HIP(hipMemcpy(table, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice)); // pageable source: returns after the copy
    c->task_orders.push_back({c->rank, c->world, table});
    c->task_macro = table;
}

void egr_trace_free(egr_context *c) {
    for (auto &o : c->task_orders) egr_dev_free(c, o.table);
    c->task_orders.clear(), c->task_macro = nullptr;
    egr_dev_free(c, c->stack_spill), egr_dev_free(c, c->cand_keys), egr_dev_free(c, c->cand_vals), egr_dev_free(c, c->hit_arena), egr_dev_free(c, c->task_last_block), egr_dev_free(c, c->task_cost), egr_dev_free(c, c->bwd_order), egr_dev_free(c, c->state), egr_dev_free(c, c->control), egr_dev_free(c, c->queues), egr_dev_free(c, c->denoise_tmp), egr_dev_free(c, c->ext_keys), egr_dev_free(c, c->ext_vals);
    for (int i = 0; i < EGR_MAX_STRANDS; i++) {
        if (c->strand_stream[i]) (void)hipStreamDestroy(c->strand_stream[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        c->strand_stream[i] = nullptr, c->ev_join[i] = nullptr;
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    c->ev_fork = nullptr;
    if (c->control_host) (void)hipHostFree(c->control_host);
    c->control_host = nullptr;
}

void egr_trace_alloc(egr_context *c) {
    uint32_t mtx = (c->width + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE, mty = (c->height + EGR_MACRO_TILE - 1) / EGR_MACRO_TILE;
    c->num_tasks_total = 4u * mtx * mty;
    hipDeviceProp_t prop;
    EGR_HIP(hipGetDeviceProperties(&prop, c->device));
    int per_cu = 0; // the forward chain is built for four waves per SIMD (faster with more waves in flight even with spills)
    EGR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_forward_chain<true, false, 1>), EGR_WAVE, 0));
    per_cu = std::max(1, std::min(32, per_cu));
    {
        int teams = 0; // the team build holds whole teams
        EGR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&teams, (k_forward_chain<true, false, EGR_TEAM>), EGR_WAVE * EGR_TEAM, 0));
        c->team_waves_per_cu = std::max(1, teams) * EGR_TEAM;
    }
    if (getenv("EGR_DEBUG_OCCUPANCY")) {
        int bwd = 0;
        EGR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bwd, k_backward_chain<1>, EGR_WAVE, 0));
        fprintf(stderr, "[egr] resident waves per CU: forward chain %d, backward chain %d\n", per_cu, bwd);
    }
    if (const char *e = getenv("EGR_WAVES_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e))); // tuning knob
    uint32_t resident = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
    c->num_slots = std::max((uint32_t)EGR_TEAM, std::min(resident, (c->num_tasks_total + EGR_TEAM - 1u) / EGR_TEAM * EGR_TEAM)); // whole teams (the workgroups of the forward chain's team build)
    // forward budget: the reference's ppll_forward_size entries x 36 B, spent on (key 4 B + value 8 B) x 64 lanes x cap per slot
    // (the leaf pairs awaiting evaluation live in LDS since the pair walk: no queue in global memory)
    double fwd_bytes = (double)c->fwd_capacity * 36.0;
    const size_t S = (size_t)c->strands; // every strand owns a full set of resident-wave scratch slots
    uint64_t cap = (uint64_t)(fwd_bytes / ((double)c->num_slots * (double)S * EGR_WAVE * 12.0)); // key 4 + value 8 bytes
    c->cand_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cap, 64), 16384) & ~7u;
    egr_dev_alloc_raw(c, (void **)&c->cand_keys, S * c->num_slots * c->cand_cap * EGR_WAVE * sizeof(float));
    egr_dev_alloc_raw(c, (void **)&c->cand_vals, S * c->num_slots * c->cand_cap * EGR_WAVE * sizeof(float2));
    // extension blocks: 1/8 of the forward byte budget on top (12 B per entry), at least 64 blocks
    c->ext_blocks_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)(fwd_bytes / 8.0 / (12.0 * EGR_EXT_BLOCK)), 64), 65536);
    egr_dev_alloc_raw(c, (void **)&c->ext_keys, (size_t)c->ext_blocks_cap * EGR_EXT_BLOCK * sizeof(float));
    egr_dev_alloc_raw(c, (void **)&c->ext_vals, (size_t)c->ext_blocks_cap * EGR_EXT_BLOCK * sizeof(float2));
    egr_dev_alloc_raw(c, (void **)&c->stack_spill, S * c->num_slots * EGR_GSTK * EGR_WAVE * sizeof(uint32_t));
    const double bwd_bytes = (double)c->bwd_capacity * 36.0; // the reference's ppll_backward_size entries x 36 B: all of it is arena
    uint64_t blocks = (uint64_t)(bwd_bytes / ((EGR_HIT_BLOCK_ROWS + 1) * EGR_WAVE * sizeof(float4)));
    c->hit_blocks_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(blocks, 64), 0x7FFFFFFFull);
    egr_dev_alloc_raw(c, (void **)&c->hit_arena, (size_t)c->hit_blocks_cap * (EGR_HIT_BLOCK_ROWS + 1) * EGR_WAVE * sizeof(float4));
    egr_dev_alloc_raw(c, (void **)&c->task_last_block, (size_t)EGR_NSTEPS * 4u * c->num_tasks_total * sizeof(uint32_t)); // (up to 16 tasks per macro tile)
    egr_dev_alloc_raw(c, (void **)&c->task_cost, (size_t)4u * c->num_tasks_total * sizeof(uint32_t));
    egr_dev_alloc_raw(c, (void **)&c->bwd_order, (size_t)4u * c->num_tasks_total * sizeof(uint32_t));
    if (const char *e = getenv("EGR_RAYS_PER_TASK")) c->rays_per_task = atoi(e);
    c->state_stride = c->num_tasks_total * EGR_WAVE;
    egr_dev_alloc_raw(c, (void **)&c->state, (size_t)F_TOTAL * c->state_stride * sizeof(float));
    EGR_HIP(hipMemset(c->state, 0, (size_t)F_TOTAL * c->state_stride * sizeof(float)));
    egr_dev_alloc_raw(c, (void **)&c->control, CW_COUNT * sizeof(uint32_t));
    EGR_HIP(hipMemset(c->control, 0, CW_COUNT * sizeof(uint32_t)));
    EGR_HIP(hipHostMalloc((void **)&c->control_host, CW_COUNT * sizeof(uint32_t)));
    egr_dev_alloc_raw(c, (void **)&c->queues, EGR_QUEUE_WORDS * S * sizeof(uint32_t));
    EGR_HIP(hipMemset(c->queues, 0, EGR_QUEUE_WORDS * S * sizeof(uint32_t)));
    if (c->strands > 1) {
        EGR_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < c->strands; i++) {
            EGR_HIP(hipStreamCreateWithFlags(&c->strand_stream[i], hipStreamNonBlocking));
            EGR_HIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
        }
    }
    egr_build_task_order(c);
}

// Team help of the next launch: on (default: a whole image's forward chain -6 % / -3 %, an under-filled rank's iteration -7 %), off, or - value -1 - on only for
// a rank of a partition with fewer than two 8x8 tiles per wave slot. With help the order of EXACT depth ties of bounce rays (0.1 % of the pixels of a 1080p frame
// hold one) varies from run to run, as it does upstream with the PPLL's insertion order; nothing else does (trace.hip: teams; forward_task.inc: total transmittance).
bool egr_team_help_on(const egr_context *c) {
    if (EGR_TEAM <= 1) return false;
    if (c->team_help >= 0) return c->team_help == 1;
    return c->world > 1 && egr_num_tasks_for_rank(c) < 2u * c->num_slots;
}

DeviceView egr_make_view(const egr_context *c) {
    DeviceView v{};
    v.width = c->width, v.height = c->height;
    v.tiles_x = (c->width + EGR_TILE - 1) / EGR_TILE, v.tiles_y = (c->height + EGR_TILE - 1) / EGR_TILE;
    v.num_pixels = (uint32_t)c->width * (uint32_t)c->height;
    v.n = c->g.count;
    v.num_nodes = c->num_wide;
    v.rank = c->rank, v.world = c->world;
    // Task size. 8x8 pixels unless the rank is under-filled: with fewer than two 8x8 tiles per resident wave (a rank of an 8-way
    // partition at 1080p) every tile starts at once and the launch lasts as long as its heaviest tile - 8x4-pixel tasks halve the
    // tiles (rank 0 of an emulated 8-way partition: 5.04 -> 4.16 ms dense-init, 3.95 -> 3.71 ms trained-like; 4x4 tasks: 4.12 / 4.12 ms,
    // the per-ray phases then run at a quarter of the lanes). A ray's candidate LIST ORDER depends on the task shape
    // (pair walk), so exactly tied depths may composite in another order than with 8x8 tasks (documented deviation (a)).
    const uint32_t tiles = egr_num_tasks_for_rank(c);
    uint32_t rpt = c->rays_per_task == 16 || c->rays_per_task == 32 || c->rays_per_task == 64 ? (uint32_t)c->rays_per_task
                   : (c->world > 1 && tiles < 2u * c->num_slots && !egr_team_help_on(c) ? 32u : 64u); // (with team help the heavy tile's walk is shared anyway, and whole 8x8 tiles keep all lanes busy in the per-ray phases: 3.08-3.12 against 3.17-3.21 ms trained-like, 3.50-3.55 against 3.48-3.51 ms dense-init per iteration of rank 0 of 8)
    v.rays_per_task = rpt, v.task_shift = rpt == 64u ? 2u : rpt == 32u ? 3u : 4u;
    v.num_tasks = tiles << (v.task_shift - 2u);
    v.task_begin = 0, v.task_count = v.num_tasks, v.queues = c->queues, v.num_strands = (uint32_t)c->strands;
    v.task_macro = c->task_macro;
    // (help changes the ORDER in which a ray's candidates enter its list, never the set; egr_team_help_on)
    v.team_help = egr_team_help_on(c) ? 1 : 0;
    v.g = c->g, v.cfg = c->cfg, v.cam = c->cam, v.fb = c->fb, v.meta = c->meta, v.stats = c->stats;
    v.wnodes = c->wnodes, v.stack_spill = c->stack_spill, v.gid_of_pos = c->vals_out, v.pos_of_gid = c->pos_of_gid, v.frame = c->frame, v.out_of_frame = c->out_of_frame, v.inst_w = c->inst_w, v.inst_m = c->inst_m, v.grad_rows = c->grad_rows, v.app = c->app, v.bsph = c->bsph;
    v.cand_keys = c->cand_keys, v.cand_vals = c->cand_vals, v.cand_cap = c->cand_cap, v.num_slots = c->num_slots;
    v.ext_keys = c->ext_keys, v.ext_vals = c->ext_vals, v.ext_blocks_cap = c->ext_blocks_cap;
    v.hit_arena = c->hit_arena, v.hit_blocks_cap = c->hit_blocks_cap, v.task_last_block = c->task_last_block, v.task_cost = c->task_cost, v.bwd_order = c->bwd_order;
    v.state = c->state, v.state_stride = c->state_stride, v.control = c->control;
    v.cube_mode = c->exact_stats ? 1 : 0;
    v.pixel_mask = c->pixel_mask;
    v.grad_overwrite = (c->grad_overwrite && !c->delta_pending) ? 1 : 0;
    return v;
}

void egr_trace_launch(egr_context *c, bool grads, bool live_fresh, hipStream_t s) {
    DeviceView v = egr_make_view(c);
    const dim3 block(EGR_WAVE);
    egr_stamp_begin(c, "prologue+live", s);
    hipLaunchKernelGGL(k_prologue, dim3(1), dim3(64), 0, s, v, grads ? 1 : 0);
    EGR_HIP(hipMemsetAsync(c->stats.num_accumulated_per_pixel, 0, sizeof(int32_t) * v.num_pixels, s)); // stats.h:25-28
    EGR_HIP(hipMemsetAsync(c->stats.num_traversed_per_pixel, 0, sizeof(int32_t) * v.num_pixels, s));
    // (skipped when the egr_update_bvh_ex(EGR_UPDATE_FUSE_LIVE) just before this launch wrote the same records from the same parameters)
    if (v.n && !live_fresh) hipLaunchKernelGGL(k_live, dim3((v.n + 255) / 256), dim3(256), 0, s, v, grads ? 1 : 0);
    egr_stamp_end(c, s);
    if (v.num_tasks) {
        // Strands: slices of the task order (whole macro tiles), each with its own queues and scratch slots, each running its
        // two chain kernels in order on its own stream. Tiles only depend on their own earlier steps, so strands never
        // synchronise with each other until the join; while one strand's kernel drains its last long tiles the next kernel of
        // another strand takes the freed wave slots.
        // (no explicit egr_set_strands: all strands when the rank has at least four tiles per wave slot; with fewer - a rank of a
        // multi-GPU partition - concurrent chains only delay each other's heaviest tiles: 4.66 / 4.85 / 4.97 ms with 1 / 2 / 3
        // strands for rank 0 of an 8-way partition, 18.7 / 18.3 / 18.0 ms for the whole image)
        // (round 5: ONE strand unless egr_set_strands asks for more. Strands paid while the chains ended in long tails; with the backward tasks costliest
        // first and today's forward chain, interleaved same-box runs of the whole image give 6.73-6.90 / 11.13 ms per iteration with one strand against
        // 6.94-6.97 / 11.33-11.38 with three - concurrent chains delay each other's tiles more than they fill tails)
        const int want = c->strands_active > 0 ? std::min(c->strands_active, c->strands) : 1;
        const int S = (v.num_tasks >= 8u * (uint32_t)want) ? want : 1;
        static const int bwd_team_env = getenv("EGR_BWD_TEAM_HELP") ? atoi(getenv("EGR_BWD_TEAM_HELP")) : -1; // (experiments: 0 / 1 force the choice)
        const bool backward_teams = bwd_team_env >= 0 ? bwd_team_env != 0 : (v.team_help == 1 || (c->world > 1 && (uint64_t)(v.num_tasks >> (v.task_shift - 2u)) < 2ull * c->num_slots)); // (egr_set_team_help(1) takes the teams of both chains: tests/test_hip_parity.py)
        if (S > 1) EGR_HIP(hipEventRecord(c->ev_fork, s));
        for (int st = 0; st < S; st++) {
            hipStream_t ls = S > 1 ? c->strand_stream[st] : s;
            if (S > 1) EGR_HIP(hipStreamWaitEvent(ls, c->ev_fork, 0));
            DeviceView w = v;
            const uint32_t groups = v.num_tasks >> v.task_shift; // tasks come in groups (one macro tile)
            w.task_begin = (uint32_t)(((uint64_t)groups * (uint64_t)st) / (uint64_t)S) << v.task_shift;
            w.task_count = ((uint32_t)(((uint64_t)groups * (uint64_t)(st + 1)) / (uint64_t)S) << v.task_shift) - w.task_begin;
            w.queues = c->queues + EGR_QUEUE_WORDS * st;
            const size_t slot0 = (size_t)st * c->num_slots;
            w.cand_keys += slot0 * c->cand_cap * EGR_WAVE, w.cand_vals += slot0 * c->cand_cap * EGR_WAVE;
            w.stack_spill += slot0 * EGR_GSTK * EGR_WAVE;
            const dim3 sgrid(std::max(1u, std::min(c->num_slots, w.task_count)));
            egr_stamp_begin(c, "forward_chain", ls);
            // (the chain exists as single-wave workgroups and as teams of EGR_TEAM waves: launches with egr_set_team_help(1) take the teams)
            auto launch_forward = [&](auto team_tag) {
                constexpr uint32_t T = (uint32_t) decltype(team_tag)::value;
                const dim3 fgrid((sgrid.x + T - 1u) / T), fblock(EGR_WAVE * T); // (num_slots is a multiple of EGR_TEAM: every wave of a team has its scratch)
                if (grads) {
                    if (v.cube_mode) hipLaunchKernelGGL((k_forward_chain<true, true, (int)T>), fgrid, fblock, 0, ls, w);
                    else hipLaunchKernelGGL((k_forward_chain<true, false, (int)T>), fgrid, fblock, 0, ls, w);
                } else {
                    if (v.cube_mode) hipLaunchKernelGGL((k_forward_chain<false, true, (int)T>), fgrid, fblock, 0, ls, w);
                    else hipLaunchKernelGGL((k_forward_chain<false, false, (int)T>), fgrid, fblock, 0, ls, w);
                }
            };
            if (v.team_help) launch_forward(std::integral_constant<int, EGR_TEAM>{});
            else launch_forward(std::integral_constant<int, 1>{});
            egr_stamp_end(c, ls);
            if (grads) {
                static const bool order_backward = !(getenv("EGR_ORDER_BACKWARD") && atoi(getenv("EGR_ORDER_BACKWARD")) == 0); // (experiments: 0 = tasks in queue order)
                w.bwd_order = order_backward ? c->bwd_order : nullptr;
                if (order_backward) hipLaunchKernelGGL(k_order_backward, dim3(8), dim3(256), 0, ls, w);
                egr_stamp_begin(c, "backward_chain", ls);
                // (under-filled ranks: the backward chain as teams whose waves without tiles take batches of their mates' bounce hits - gradients are
                // atomic adds, so no result depends on who sends them; a whole image keeps single-wave workgroups: a team's LDS is only released with its last wave)
                if (backward_teams) hipLaunchKernelGGL(k_backward_chain<EGR_BWD_TEAM>, dim3((sgrid.x + EGR_BWD_TEAM - 1u) / EGR_BWD_TEAM), dim3(EGR_WAVE * EGR_BWD_TEAM), 0, ls, w);
                else hipLaunchKernelGGL(k_backward_chain<1>, sgrid, block, 0, ls, w);
                egr_stamp_end(c, ls);
            } else {
                egr_stamp_begin(c, "write_outputs", ls);
                hipLaunchKernelGGL(k_finish, dim3(std::max(1u, std::min(w.task_count, 65535u))), block, 0, ls, w);
                egr_stamp_end(c, ls);
            }
            if (S > 1) {
                EGR_HIP(hipEventRecord(c->ev_join[st], ls));
                EGR_HIP(hipStreamWaitEvent(s, c->ev_join[st], 0));
            }
        }
    }
    if (grads && v.n && (v.num_tasks || c->grad_overwrite)) { // (a rank without tiles still owes its per-launch buffer a row of zeros)
        egr_stamp_begin(c, "backward_grad_gather", s);
        // (per-launch buffer: the FIRST grad launch after the caller consumed the buffer stores, any further one before the next
        // egr_grad_delta_consumed adds - two launches before a fold never drop the first one's gradients)
        if (v.grad_overwrite) hipLaunchKernelGGL(k_grad_gather<true>, dim3((v.n + 255u) / 256u), dim3(256), 0, s, v);
        else hipLaunchKernelGGL(k_grad_gather<false>, dim3((v.n + 255u) / 256u), dim3(256), 0, s, v);
        egr_stamp_end(c, s);
        if (c->grad_overwrite) c->delta_pending = true;
    }
    hipLaunchKernelGGL(k_epilogue, dim3(1), dim3(64), 0, s, v, grads ? 1 : 0);
}

void egr_export_step_hits(egr_context *c, int32_t *host_out, hipStream_t s) {
    DeviceView v = egr_make_view(c);
    const size_t bytes = (size_t)EGR_NSTEPS * v.num_pixels * sizeof(int32_t);
    struct DevBuf { // freed on every way out (an EGR_HIP below may throw)
        int32_t *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } dev;
    EGR_HIP(hipMalloc((void **)&dev.p, bytes));
    EGR_HIP(hipMemsetAsync(dev.p, 0, bytes, s));
    if (v.num_tasks) hipLaunchKernelGGL(k_export_step_hits, dim3(std::min(v.num_tasks, 65535u)), dim3(EGR_WAVE), 0, s, v, dev.p);
    EGR_HIP(hipMemcpyAsync(host_out, dev.p, bytes, hipMemcpyDeviceToHost, s));
    EGR_HIP(hipStreamSynchronize(s));
}

void egr_export_hit_hash(egr_context *c, uint64_t *host_out, hipStream_t s) {
    DeviceView v = egr_make_view(c);
    const size_t bytes = (size_t)EGR_NSTEPS * v.num_pixels * sizeof(uint64_t);
    struct DevBuf {
        unsigned long long *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } dev;
    EGR_HIP(hipMalloc((void **)&dev.p, bytes));
    EGR_HIP(hipMemsetAsync(dev.p, 0, bytes, s));
    if (v.num_tasks) hipLaunchKernelGGL(k_export_hit_hash, dim3(std::min(v.num_tasks, 65535u)), dim3(EGR_WAVE), 0, s, v, dev.p);
    EGR_HIP(hipMemcpyAsync(host_out, dev.p, bytes, hipMemcpyDeviceToHost, s));
    EGR_HIP(hipStreamSynchronize(s));
}

void egr_set_camera_launch(egr_context *c, const float *R, const float *centre, float fov, float znear, float zfar, hipStream_t s) {
    hipLaunchKernelGGL(k_set_camera, dim3(1), dim3(64), 0, s, c->cam, R, centre, fov, znear, zfar);
}

void egr_upload_targets(egr_context *c, const float *const chw[6], hipStream_t s) {
    DeviceView v = egr_make_view(c);
    v.pixel_mask = nullptr; // (targets of every pixel of the rank's tiles, whatever a debug mask says)
    TargetImages t;
    for (int b = 0; b < 6; b++) t.chw[b] = chw[b];
    bool flat = c->world == 1 && v.num_pixels % 4 == 0 && v.num_pixels != 0;
    const float *const fbt[6] = {v.fb.target_diffuse, v.fb.target_specular, v.fb.target_depth, v.fb.target_normal, v.fb.target_roughness, v.fb.target_f0};
    for (int b = 0; b < 6; b++) flat = flat && ((reinterpret_cast<uintptr_t>(chw[b]) | reinterpret_cast<uintptr_t>(fbt[b])) & 15u) == 0;
    if (flat)
        hipLaunchKernelGGL(k_upload_targets_flat, dim3((unsigned)std::min<size_t>((v.num_pixels / 4 + 255) / 256, 4096)), dim3(256), 0, s, v, t);
    else if (v.num_tasks)
        hipLaunchKernelGGL(k_upload_targets, dim3(std::min(v.num_tasks, 65535u)), dim3(EGR_WAVE), 0, s, v, t);
}

void egr_copy_final_to_denoised(egr_context *c, hipStream_t s) {
    size_t n = (size_t)c->width * c->height * 3;
    hipLaunchKernelGGL(k_copy3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->fb.output_final, c->fb.output_denoised, n);
}
