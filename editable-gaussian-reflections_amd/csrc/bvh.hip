// Instance transforms + LBVH build/refit for gfx950. Replaces the reference's K1/K2:
//   _populateBVH / create_transform_matrix   (optix/bvh_wrapper.cu:9-59)
//   OptiX TLAS build / refit                 (optix/bvh_wrapper.h:32-59,118-157; closed source)
//
// Design (MI355X-first, not a translation of OptiX instancing):
//   * the incoherent (bounce) walk is bound by cache-line fills, not instructions: every divergent 16-B gather pulls
//     a whole 128-B line (PMC, profiles/r1: L1 hit 56 %, L2 hit 63 %, ~10x re-fetch). So a node IS one line:
//     8 children x 16 B = 128 B, aligned. A child slot = box quantised to 16 bits per coordinate in a per-build frame
//     (conservative: one extra cell each side; cells 0 / 65535 decode to -inf / +inf so boxes that drift out of the
//     frame between rebuilds stay correct) + 32-bit link (leaf: record index, internal: child node, or empty);
//   * topology: binary LBVH from 63-bit Morton codes of the Gaussian means (Karras 2012), collapsed to 8-wide nodes
//     level by level (always expanding the child with the most leaves); built only on rebuild;
//   * the per-iteration refit is atomics-free and fence-free: wide nodes of one level occupy a contiguous index
//     range and are refitted deepest level first, one launch per level (kernel boundaries provide the ordering the
//     non-coherent per-XCD L2s would otherwise need agent-scope fences for).
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "egr_device.hpp"
#include "egr_internal.hpp"

namespace {

constexpr int BS = 256;
inline int nblk(uint64_t n, int bs = BS) { return (int)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------------------------
// K1: per-Gaussian instance record. M = [R diag(exp(s) * sigma * g) | mean], W = M^-1 (analytic).
// mask (bvh_wrapper.cu:55): sigma > 0 && any(size > 0); masked-out instances get an empty box (lo > hi) and a W
// record that can never pass the cube test (they keep their leaf slot, so a walk may still look at them).
// ---------------------------------------------------------------------------------------------------------
// Records are stored at the gaussian's MORTON-SORTED position (pos_of_gid), i.e. in leaf order: rays that are close
// in space then read neighbouring 48-B records (same / adjacent cache lines) instead of lines scattered over the
// whole array - the scene's own order is arbitrary (the reference never sorts its clouds).
// `live` (egr_update_bvh_ex with EGR_UPDATE_FUSE_LIVE): also write what k_live (trace.hip) writes at every launch - the activated appearance
// and the (f0.z, roughness, opacity, sigma) quarter of the test record - so that the raytrace that follows immediately skips that pass
// over the cloud (the records' lines are being written here anyway).
__global__ void __launch_bounds__(BS) k_instances(uint32_t n, egr_gaussians g, egr_config cfg, const uint32_t *__restrict__ pos_of_gid,
                                                  float4 *__restrict__ inst_w, float4 *__restrict__ inst_m, float *__restrict__ aabb, int cube,
                                                  float4 *__restrict__ app, int live, float4 *__restrict__ bsph) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = pos_of_gid ? pos_of_gid[i] : i;
    const float alpha_threshold = *cfg.alpha_threshold, exp_power = *cfg.exp_power, gsf = *cfg.global_scale_factor;
    float opacity = sigmoid_act(g.opacity[i]);
    float sf = compute_scaling_factor(opacity, alpha_threshold, exp_power);
    f3 sizes = mk3(expf(g.scale[3 * i]), expf(g.scale[3 * i + 1]), expf(g.scale[3 * i + 2]));
    sizes = (sizes * sf) * gsf; // bvh_wrapper.cu:49-53
    bool visible = sf > 0.0f && (sizes.x > 0.0f || sizes.y > 0.0f || sizes.z > 0.0f);
    const float4 q4 = reinterpret_cast<const float4 *>(g.rotation)[i];
    float r = q4.x, x = q4.y, y = q4.z, z = q4.w;
    float norm = sqrtf(r * r + x * x + y * y + z * z); // bvh_wrapper.cu:18-22
    r /= norm, x /= norm, y /= norm, z /= norm;
    float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                     {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                     {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float s[3] = {sizes.x, sizes.y, sizes.z};
    float m[3] = {g.mean[3 * i], g.mean[3 * i + 1], g.mean[3 * i + 2]};
    float inv[3] = {1.0f / s[0], 1.0f / s[1], 1.0f / s[2]};
    float lo[3], hi[3];
    float4 Wr[3];
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float M0 = s[0] * R[a][0], M1 = s[1] * R[a][1], M2 = s[2] * R[a][2];
        inst_m[4 * slot + a] = make_float4(M0, M1, M2, expf(g.scale[3 * i + a])); // .w: activated scale (backward_pass.cu:170)
        float W0 = R[0][a] * inv[a], W1 = R[1][a] * inv[a], W2 = R[2][a] * inv[a];
        Wr[a] = make_float4(W0, W1, W2, -(W0 * m[0] + W1 * m[1] + W2 * m[2]));
        finite = finite && isfinite(W0 + W1 + W2 + Wr[a].w);
        // Box of the ELLIPSOID M * (unit sphere): half-extent = |row of M|_2 (not the cube's |row|_1). Accepted hits have
        // their response point inside it (see the three-segment walk in trace.hip); padded against fp32 rounding.
        // Exact-statistics mode (egr_set_exact_stats): the box of the instance CUBE M * [-1,1]^3, what OptiX's TLAS bounds.
        float ext = cube ? fabsf(M0) + fabsf(M1) + fabsf(M2) : sqrtf(M0 * M0 + M1 * M1 + M2 * M2);
        ext = ext * 1.0001f + 4e-7f * (fabsf(m[a]) + ext);
        lo[a] = m[a] - ext;
        hi[a] = m[a] + ext;
        finite = finite && (lo[a] <= hi[a]) && isfinite(lo[a]) && isfinite(hi[a]);
    }
    inst_m[4 * slot + 3] = q4; // raw quaternion for the normalisation backward (activations.cu:66-73)
    const bool usable = visible && finite; // NaN / inf parameters can never be hit
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // unusable: object-space origin (2,2,2), zero direction -> outside the unit cube for every ray
        inst_w[4 * slot + a] = usable ? Wr[a] : make_float4(0.f, 0.f, 0.f, 2.f); // 64-B record: rows 0-2 = W, row 3 = live quarter (k_live)
        aabb[6 * i + a] = usable ? lo[a] : 3.0e38f;
        aabb[6 * i + 3 + a] = usable ? hi[a] : -3.0e38f;
    }
    {
        // bounding SPHERE of what the box bounds (centre, squared radius; record order): the ellipsoid's largest semi-axis, in exact-statistics mode
        // the cube's half diagonal. Primary tiles test it per ray before a (ray, gaussian) pair is evaluated at all (forward_task.inc); padded like the
        // box, the ray-dependent part of the slack is added there. An unusable gaussian has a radius no ray meets.
        float rad = cube ? sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) : fmaxf(s[0], fmaxf(s[1], s[2]));
        rad = rad * 1.0001f + 4e-7f * (fabsf(m[0]) + fabsf(m[1]) + fabsf(m[2]) + rad);
        bsph[slot] = usable && isfinite(rad) ? make_float4(m[0], m[1], m[2], rad * rad) : make_float4(0.f, 0.f, 0.f, -3.0e38f);
    }
    if (live) { // the same expressions as k_live (utils/helpers.cu:10-33)
        app[2 * (size_t)slot] = make_float4(relu_act(g.rgb[3 * i]), relu_act(g.rgb[3 * i + 1]), relu_act(g.rgb[3 * i + 2]), g.normal[3 * i]);
        app[2 * (size_t)slot + 1] = make_float4(g.normal[3 * i + 1], g.normal[3 * i + 2], clip01_act(g.f0[3 * i]), clip01_act(g.f0[3 * i + 1]));
        inst_w[4 * (size_t)slot + 3] = make_float4(clip01_act(g.f0[3 * i + 2]), clip01_act(g.roughness[i]), opacity, sf);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Frame (bounds of all usable boxes) and Morton codes of the means
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float f_unordered_host(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__global__ void k_bounds_init(uint32_t *b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(BS) k_bounds(uint32_t n, const float *__restrict__ aabb, uint32_t *__restrict__ b) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool ok = false;
    if (i < n) {
#pragma unroll
        for (int a = 0; a < 3; a++) lo[a] = aabb[6 * i + a], hi[a] = aabb[6 * i + 3 + a];
        ok = lo[0] <= hi[0];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        uint32_t l = ok ? f_ordered(lo[a]) : 0xFFFFFFFFu, h = ok ? f_ordered(hi[a]) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            l = min(l, (uint32_t)__shfl_xor((int)l, off));
            h = max(h, (uint32_t)__shfl_xor((int)h, off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&b[a], l);
            atomicMax(&b[3 + a], h);
        }
    }
}
__device__ __forceinline__ uint64_t spread21(uint32_t v) { // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// Sort key of a gaussian: 63-bit Morton code of its mean in the build frame (21 bits per axis).
__global__ void __launch_bounds__(BS) k_morton(uint32_t n, const float *__restrict__ mean, BvhFrame fr, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    const float fo[3] = {fr.ox, fr.oy, fr.oz}, fs[3] = {fr.sx, fr.sy, fr.sz};
    uint32_t q[3]; // 21-bit fixed point in [0,1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float v = mean[3 * i + a];
        float u = isfinite(v) ? (v - fo[a]) * fs[a] * (1.0f / 65536.0f) : 0.0f; // [0,1) inside the frame
        q[a] = (uint32_t)fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}
__global__ void __launch_bounds__(BS) k_inverse_perm(uint32_t n, const uint32_t *__restrict__ gid_of_pos, uint32_t *__restrict__ pos_of_gid) {
    uint32_t p = blockIdx.x * BS + threadIdx.x;
    if (p < n) pos_of_gid[gid_of_pos[p]] = p;
}

// ---------------------------------------------------------------------------------------------------------
// Karras 2012 topology over the sorted gaussians. Id space: internal i in [0,n-2] -> i, leaf j in [0,n-1] -> (n-1)+j.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int kdelta(const uint64_t *__restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(i ^ j);
    return __clzll((long long)(a ^ b));
}
__global__ void __launch_bounds__(BS) k_karras(int n, const uint64_t *__restrict__ keys, int32_t *__restrict__ left,
                                               int32_t *__restrict__ right, int32_t *__restrict__ parent,
                                               uint32_t *__restrict__ first, uint32_t *__restrict__ last) {
    int i = blockIdx.x * BS + threadIdx.x;
    if (i >= n - 1) return;
    int d = (kdelta(keys, n, i, i + 1) - kdelta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = kdelta(keys, n, i, i - d);
    int lmax = 2;
    while (kdelta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (kdelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = kdelta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (kdelta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc = (lo == gamma) ? (n - 1) + gamma : gamma;
    int rc = (hi == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    parent[lc] = i;
    parent[rc] = i;
    first[i] = (uint32_t)lo;
    last[i] = (uint32_t)hi;
    if (i == 0) parent[0] = -1;
}

// ---------------------------------------------------------------------------------------------------------
// SAH-optimal collapse, part 1 (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through compressed wide
// BVHs", sec. 4.2, restated for one gaussian per leaf): the walk's cost is the number of 128-B wide nodes a ray enters, and a ray
// enters a node with probability ~ its surface area, so the best 8-wide tree over a given binary tree minimises the SUM OF THE AREAS
// OF ITS WIDE NODES (every leaf costs the same wherever it hangs). Dynamic programme over the binary tree, bottom-up:
//   T(n, i) = least total area of the wide nodes inside subtree n when n may occupy at most i child slots of its parent's wide node
//   leaf:      T = 0
//   internal:  T(n, 1) = area(n) + D(n, 8)                       (n becomes a wide node, its 8 slots go to its two children)
//              T(n, i) = min(D(n, i), T(n, i - 1)),  i = 2..7    (n dissolves into the parent's node)
//              D(n, j) = min over k = 1..j-1 of T(left, k) + T(right, j - k)
// One thread per leaf climbs; the second arrival at a node owns it (atomic counter). The tree is spread over all XCDs whose L2s
// are not coherent (cdna guide, Guideline 16): every payload word is written and read with agent-scope atomic accesses (sc1,
// write-through / cache-bypassing), the writer drains its stores before the counter RMW, which itself is acq_rel at agent scope.
// Record per binary internal node: box lo[3], hi[3], T[1..7] (16 floats, 3 unused). Runs once per rebuild.
// ---------------------------------------------------------------------------------------------------------
#define EGR_DP_STRIDE 16
__device__ __forceinline__ float ld_agent(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// D(n, j) and the split that attains it, from the children's T rows (index 0 unused; a leaf's row is all zero)
__device__ __forceinline__ float dp_distribute(const float (&Tl)[8], const float (&Tr)[8], int j, int &best_k) {
    float best = 3.0e38f;
    best_k = 1;
    for (int k = 1; k < j; k++) {
        const float c = Tl[min(k, 7)] + Tr[min(j - k, 7)];
        if (c < best) best = c, best_k = k;
    }
    return best;
}
__global__ void __launch_bounds__(BS) k_sah_bottom_up(int n, const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                                      const int32_t *__restrict__ parent, const float *__restrict__ aabb,
                                                      const uint32_t *__restrict__ gid_of_pos, uint32_t *flags, float *dp) {
    const int j = blockIdx.x * BS + threadIdx.x;
    if (j >= n) return;
    int cur = parent[(n - 1) + j];
    while (cur >= 0) {
        if (__hip_atomic_fetch_add(&flags[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0u) return; // the sibling subtree is not done yet
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        float Tc[2][8];
        const int ch[2] = {left[cur], right[cur]};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (ch[c] >= n - 1) { // leaf: the gaussian's box (plain loads: written by an earlier kernel), no wide nodes below
                const float *bx = aabb + 6 * (size_t)gid_of_pos[ch[c] - (n - 1)];
#pragma unroll
                for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], bx[a]), hi[a] = fmaxf(hi[a], bx[3 + a]);
#pragma unroll
                for (int i = 0; i < 8; i++) Tc[c][i] = 0.0f;
            } else {
                const float *r = dp + (size_t)ch[c] * EGR_DP_STRIDE;
#pragma unroll
                for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], ld_agent(r + a)), hi[a] = fmaxf(hi[a], ld_agent(r + 3 + a));
                Tc[c][0] = 0.0f;
#pragma unroll
                for (int i = 1; i < 8; i++) Tc[c][i] = ld_agent(r + 5 + i);
            }
        }
        const float dx = fmaxf(hi[0] - lo[0], 0.0f), dy = fmaxf(hi[1] - lo[1], 0.0f), dz = fmaxf(hi[2] - lo[2], 0.0f);
        const float area = (lo[0] <= hi[0]) ? dx * dy + dy * dz + dz * dx : 0.0f; // (half the surface area; empty box: unusable gaussians only)
        float T[8];
        int k_unused;
        T[0] = 0.0f;
        T[1] = area + dp_distribute(Tc[0], Tc[1], 8, k_unused);
        for (int i = 2; i < 8; i++) T[i] = fminf(dp_distribute(Tc[0], Tc[1], i, k_unused), T[i - 1]);
        float *w = dp + (size_t)cur * EGR_DP_STRIDE;
#pragma unroll
        for (int a = 0; a < 3; a++) st_agent(w + a, lo[a]), st_agent(w + 3 + a, hi[a]);
#pragma unroll
        for (int i = 1; i < 8; i++) st_agent(w + 5 + i, T[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the record is out before the parent's counter moves
        cur = parent[cur];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Collapse of the binary tree into 8-wide nodes, one level of the wide tree per launch.
// frontier_in: binary internal ids that ARE wide nodes of this level (their wide index is wide_of[id]).
// Children = the binary node's two children, then repeatedly the internal child with the most leaves is replaced by
// its own two children until there are 8 (or only leaves). Internal children become wide nodes of the next level.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BS) k_collapse_level(int n, uint32_t count, const uint32_t *__restrict__ frontier_in,
                                                       const int32_t *__restrict__ left, const int32_t *__restrict__ right,
                                                       const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                                       uint32_t *__restrict__ wide_of, uint32_t *__restrict__ counters, // [0] = #wide nodes, [1] = next frontier size
                                                       uint32_t *__restrict__ frontier_out, uint4 *__restrict__ wnodes, const float *__restrict__ dp) {
    uint32_t t = blockIdx.x * BS + threadIdx.x;
    if (t >= count) return;
    const int b = (int)frontier_in[t];
    const uint32_t w = wide_of[b];
    int child[EGR_WIDTH];
    int nchild = 2;
    child[0] = left[b], child[1] = right[b];
    (void)first, (void)last;
    // SAH-optimal collapse, part 2: replay the dynamic programme's decisions top-down. Every entry carries the number of slots the
    // optimum grants its subtree; an internal entry with more than one slot is replaced by its two children with the split that
    // attains D(node, slots) (kept in Morton order) - unless the extra slots do not pay (T(node, i) == T(node, i - 1)).
    {
        int budget[EGR_WIDTH];
        auto row = [&](int id, float (&T)[8]) {
            T[0] = 0.0f;
            for (int i = 1; i < 8; i++) T[i] = id >= n - 1 ? 0.0f : dp[(size_t)id * EGR_DP_STRIDE + 5 + i];
        };
        {
            float Tl[8], Tr[8];
            row(child[0], Tl), row(child[1], Tr);
            int k;
            dp_distribute(Tl, Tr, EGR_WIDTH, k);
            budget[0] = k, budget[1] = EGR_WIDTH - k;
        }
        for (int at = 0; at < nchild;) {
            const int id = child[at];
            if (id >= n - 1) { at++; continue; } // a leaf takes one slot
            float T[8], Tl[8], Tr[8];
            row(id, T), row(left[id], Tl), row(right[id], Tr);
            int b = min(budget[at], 7), k = 1;
            // T(id, b) = min(D(id, b), T(id, b - 1)): slots that buy nothing stay empty; at equal cost the flatter tree wins (fewer
            // nodes; all-zero areas - coincident gaussians - would otherwise keep the binary tree's depth)
            while (b >= 2 && !(dp_distribute(Tl, Tr, b, k) <= T[b - 1])) b--;
            if (b == 1) { budget[at] = 1; at++; continue; } // stays a wide node of its own
            for (int q = nchild; q > at + 1; q--) child[q] = child[q - 1], budget[q] = budget[q - 1];
            child[at] = left[id], budget[at] = k;
            child[at + 1] = right[id], budget[at + 1] = b - k;
            nchild++; // (sum of budgets <= 8 and every entry holds >= 1: never more than 8 entries)
        }
    }
    for (int k = 0; k < EGR_WIDTH; k++) {
        uint32_t link = EGR_EMPTY_SLOT;
        if (k < nchild) {
            const int id = child[k];
            if (id >= n - 1) {
                link = EGR_LEAF_FLAG | (uint32_t)(id - (n - 1)); // leaf: sorted position of the gaussian
            } else {
                const uint32_t idx = atomicAdd(&counters[0], 1u);
                wide_of[id] = idx;
                frontier_out[atomicAdd(&counters[1], 1u)] = (uint32_t)id;
                link = idx;
            }
        }
        wnodes[(size_t)w * EGR_WIDTH + k] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, link); // empty box until refit
    }
}

// ---- 16-bit box quantisation in the build frame. u = (x - origin) * scale + 2 (cells); lo rounds down one extra
// cell, hi up one extra cell; 0 / 65535 are the out-of-frame sentinels (decoded as -inf / +inf by the traversal).
__device__ __forceinline__ uint32_t quant_lo(float x, float o, float s) {
    float u = (x - o) * s + 2.0f;
    return (u >= 2.0f) ? (uint32_t)fminf(floorf(u) - 1.0f, 65534.0f) : 0u;
}
__device__ __forceinline__ uint32_t quant_hi(float x, float o, float s) {
    float u = (x - o) * s + 2.0f;
    return (u <= 65533.0f) ? (uint32_t)fmaxf(ceilf(u) + 1.0f, 1.0f) : 65535u;
}
__device__ __forceinline__ void unpack_box(uint4 q, uint32_t lo[3], uint32_t hi[3]) {
    lo[0] = q.x & 0xFFFFu, lo[1] = q.x >> 16, lo[2] = q.y & 0xFFFFu, hi[0] = q.y >> 16, hi[1] = q.z & 0xFFFFu, hi[2] = q.z >> 16;
}
__device__ __forceinline__ uint4 pack_box(const uint32_t lo[3], const uint32_t hi[3], uint32_t link) {
    return make_uint4(lo[0] | (lo[1] << 16), lo[2] | (hi[0] << 16), hi[1] | (hi[2] << 16), link);
}
// One level of the wide tree: every child slot's box. Leaf slot: the gaussian's (padded) world box. Internal slot: the
// union of the child node's 8 slot boxes - the child lives in a deeper level that an earlier launch already refitted.
__global__ void __launch_bounds__(BS) k_refit_wide_level(uint32_t begin, uint32_t end, const float *__restrict__ aabb,
                                                         const uint32_t *__restrict__ gid_of_pos, BvhFrame fr, uint4 *wnodes, uint32_t *__restrict__ out_of_frame) {
    const uint32_t slot = blockIdx.x * BS + threadIdx.x; // one thread per child slot
    const uint32_t w = begin + slot / EGR_WIDTH, k = slot % EGR_WIDTH;
    if (w >= end) return;
    const uint32_t link = wnodes[(size_t)w * EGR_WIDTH + k].w;
    uint32_t lo[3] = {65535u, 65535u, 65535u}, hi[3] = {0u, 0u, 0u};
    if (link == EGR_EMPTY_SLOT) {
    } else if (link & EGR_LEAF_FLAG) {
        const uint32_t gid = gid_of_pos[link & ~EGR_LEAF_FLAG];
        const float *bx = aabb + 6 * (size_t)gid;
        if (bx[0] <= bx[3]) {
            const float fo[3] = {fr.ox, fr.oy, fr.oz}, fs[3] = {fr.sx, fr.sy, fr.sz};
#pragma unroll
            for (int a = 0; a < 3; a++) lo[a] = quant_lo(bx[a], fo[a], fs[a]), hi[a] = quant_hi(bx[3 + a], fo[a], fs[a]);
            // a box that left the build frame carries the -inf / +inf sentinel cells: the traversal then takes its slower decode
            if (lo[0] == 0u || lo[1] == 0u || lo[2] == 0u || hi[0] == 65535u || hi[1] == 65535u || hi[2] == 65535u) atomicOr(out_of_frame, 1u);
        }
    } else {
        const uint4 *ch = wnodes + (size_t)link * EGR_WIDTH;
#pragma unroll
        for (int c = 0; c < EGR_WIDTH; c++) {
            uint32_t l[3], h[3];
            unpack_box(ch[c], l, h);
#pragma unroll
            for (int a = 0; a < 3; a++) lo[a] = min(lo[a], l[a]), hi[a] = max(hi[a], h[a]);
        }
    }
    wnodes[(size_t)w * EGR_WIDTH + k] = pack_box(lo, hi, link);
}


} // namespace

template <class T> static void dalloc(egr_context *c, T *&p, size_t count) {
    egr_dev_free(c, p);
    egr_dev_alloc(c, p, count);
}

void egr_bvh_free(egr_context *c) {
    egr_dev_free(c, c->wnodes), egr_dev_free(c, c->pos_of_gid), egr_dev_free(c, c->inst_w), egr_dev_free(c, c->inst_m), egr_dev_free(c, c->bsph), egr_dev_free(c, c->app), egr_dev_free(c, c->aabb), egr_dev_free(c, c->grad_rows);
    egr_dev_free(c, c->sort_tmp), egr_dev_free(c, c->keys_in), egr_dev_free(c, c->keys_out), egr_dev_free(c, c->vals_in), egr_dev_free(c, c->vals_out);
    egr_dev_free(c, c->k_left), egr_dev_free(c, c->k_right), egr_dev_free(c, c->k_parent), egr_dev_free(c, c->k_first), egr_dev_free(c, c->k_last), egr_dev_free(c, c->k_dp), egr_dev_free(c, c->k_flags), egr_dev_free(c, c->wide_of), egr_dev_free(c, c->scratch_u32), egr_dev_free(c, c->out_of_frame);
    c->n_alloc = 0;
    c->n_built = 0;
    c->bvh_valid = false;
}

void egr_bvh_reserve(egr_context *c, uint32_t n) {
    if (n <= c->n_alloc && c->wnodes) return;
    uint32_t cap = std::max<uint32_t>(n + n / 8, 256); // head-room: the reference grows by +75k far-field points
    dalloc(c, c->wnodes, (size_t)cap * EGR_WIDTH + 64);   // <= n-1 wide nodes (one per binary internal node, usually ~n/5)
    dalloc(c, c->pos_of_gid, cap);
    dalloc(c, c->inst_w, 4 * (size_t)cap);
    dalloc(c, c->inst_m, 4 * (size_t)cap);
    dalloc(c, c->bsph, (size_t)cap);
    dalloc(c, c->grad_rows, 32 * (size_t)cap);
    EGR_HIP(hipMemset(c->grad_rows, 0, 32 * (size_t)cap * sizeof(float)));
    dalloc(c, c->app, 3 * (size_t)cap);
    dalloc(c, c->aabb, 6 * (size_t)cap);
    dalloc(c, c->keys_in, cap), dalloc(c, c->keys_out, cap), dalloc(c, c->vals_in, cap), dalloc(c, c->vals_out, cap);
    dalloc(c, c->k_left, cap), dalloc(c, c->k_right, cap), dalloc(c, c->k_parent, 2 * (size_t)cap);
    dalloc(c, c->k_first, cap), dalloc(c, c->k_last, cap), dalloc(c, c->wide_of, cap);
    dalloc(c, c->k_dp, (size_t)cap * EGR_DP_STRIDE), dalloc(c, c->k_flags, cap);
    if (!c->scratch_u32) dalloc(c, c->scratch_u32, 64);
    if (!c->out_of_frame) {
        dalloc(c, c->out_of_frame, 1);
        EGR_HIP(hipMemset(c->out_of_frame, 0, sizeof(uint32_t)));
    }
    size_t bytes = 0;
    EGR_HIP(rocprim::radix_sort_pairs(nullptr, bytes, c->keys_in, c->keys_out, c->vals_in, c->vals_out, (size_t)cap, 0, 63, 0));
    egr_dev_free(c, c->sort_tmp);
    egr_dev_alloc_raw(c, &c->sort_tmp, bytes);
    c->sort_tmp_bytes = bytes;
    c->n_alloc = cap;
    c->bvh_valid = false;
}

static void refit_boxes(egr_context *c, hipStream_t s) {
    EGR_HIP(hipMemsetAsync(c->out_of_frame, 0, sizeof(uint32_t), s));
    // level_start[L]..level_start[L+1] = wide nodes of level L (root = level 0). Deepest level first.
    for (int L = (int)c->level_start.size() - 2; L >= 0; L--) {
        const uint32_t b = c->level_start[L], e = c->level_start[L + 1];
        if (e > b)
            hipLaunchKernelGGL(k_refit_wide_level, dim3(nblk((uint64_t)(e - b) * EGR_WIDTH)), dim3(BS), 0, s, b, e, c->aabb, c->vals_out, c->frame,
                               c->wnodes, c->out_of_frame);
    }
}

void egr_bvh_rebuild(egr_context *c, hipStream_t s) {
    const uint32_t n = c->g.count;
    // the pair walk packs (ray, record) and (ray, wide node) into 32 bits: 6 + 26 (trace.hip). 67M gaussians is beyond any scene this
    // path has seen (50 GB of records); fail loudly rather than wrap.
    if (n >= (1u << 26)) throw EgrCheck{hipErrorInvalidValue, "rebuild_bvh: more than 2^26 - 1 gaussians are not supported (record index is packed into 26 bits)"};
    egr_bvh_reserve(c, n);
    c->live_fresh = false; // the records move
    c->n_built = n;
    c->num_wide = 0;
    c->max_depth = 0;
    c->level_start.assign(1, 0);
    c->frame = BvhFrame{0.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    c->boxes_are_cubes = c->exact_stats;
    if (n == 0) {
        c->bvh_valid = true;
        return;
    }
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)nullptr, c->inst_w, c->inst_m, c->aabb, c->exact_stats ? 1 : 0, c->app, 0, c->bsph); // boxes for the frame
    uint32_t *bounds = c->scratch_u32, *counters = c->scratch_u32 + 16;
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, s, bounds);
    hipLaunchKernelGGL(k_bounds, dim3(nblk(n)), dim3(BS), 0, s, n, c->aabb, bounds);
    uint32_t hb[6];
    EGR_HIP(hipMemcpyAsync(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, s));
    EGR_HIP(hipStreamSynchronize(s));
    {
        float lo[3], hi[3];
        bool any = hb[0] != 0xFFFFFFFFu;
        for (int a = 0; a < 3; a++) {
            lo[a] = any ? f_unordered_host(hb[a]) : 0.0f;
            hi[a] = any ? f_unordered_host(hb[3 + a]) : 1.0f;
            float ext = std::max(hi[a] - lo[a], 1e-6f);
            lo[a] -= 0.05f * ext, hi[a] += 0.05f * ext; // head-room for motion between rebuilds
        }
        c->frame.ox = lo[0], c->frame.oy = lo[1], c->frame.oz = lo[2];
        c->frame.sx = 65530.0f / (hi[0] - lo[0]), c->frame.sy = 65530.0f / (hi[1] - lo[1]), c->frame.sz = 65530.0f / (hi[2] - lo[2]);
    }
    hipLaunchKernelGGL(k_morton, dim3(nblk(n)), dim3(BS), 0, s, n, c->g.mean, c->frame, c->keys_in, c->vals_in);
    size_t bytes = c->sort_tmp_bytes;
    EGR_HIP(rocprim::radix_sort_pairs(c->sort_tmp, bytes, c->keys_in, c->keys_out, c->vals_in, c->vals_out, (size_t)n, 0, 63, s));
    hipLaunchKernelGGL(k_inverse_perm, dim3(nblk(n)), dim3(BS), 0, s, n, c->vals_out, c->pos_of_gid);
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)c->pos_of_gid, c->inst_w, c->inst_m, c->aabb, c->exact_stats ? 1 : 0, c->app, 0, c->bsph); // records in leaf order
    if (n == 1) { // a single leaf under a one-child root
        uint4 root[EGR_WIDTH];
        for (int k = 0; k < EGR_WIDTH; k++) root[k] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, k == 0 ? (EGR_LEAF_FLAG | 0u) : EGR_EMPTY_SLOT);
        EGR_HIP(hipMemcpyAsync(c->wnodes, root, sizeof(root), hipMemcpyHostToDevice, s));
        EGR_HIP(hipStreamSynchronize(s));
        c->num_wide = 1;
        c->level_start = {0, 1};
    } else {
        hipLaunchKernelGGL(k_karras, dim3(nblk(n - 1)), dim3(BS), 0, s, (int)n, c->keys_out, c->k_left, c->k_right, c->k_parent, c->k_first,
                           c->k_last);
        EGR_HIP(hipMemsetAsync(c->k_flags, 0, (size_t)n * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_sah_bottom_up, dim3(nblk(n)), dim3(BS), 0, s, (int)n, c->k_left, c->k_right, c->k_parent, c->aabb, c->vals_out, c->k_flags, c->k_dp);
        // level-by-level collapse; frontiers ping-pong in keys_in (free after the sort)
        uint32_t *fr0 = reinterpret_cast<uint32_t *>(c->keys_in), *fr1 = fr0 + c->n_alloc;
        uint32_t init[4] = {1u, 0u, 0u, 0u}; // wide node 0 = binary root
        uint32_t zero = 0;
        EGR_HIP(hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, s));
        EGR_HIP(hipMemcpyAsync(fr0, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, s));     // frontier = {binary root}
        EGR_HIP(hipMemcpyAsync(c->wide_of, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, s)); // wide_of[root] = 0
        uint32_t count = 1, nw = 1;
        c->level_start = {0, 1};
        while (count > 0) {
            hipLaunchKernelGGL(k_collapse_level, dim3(nblk(count)), dim3(BS), 0, s, (int)n, count, fr0, c->k_left, c->k_right, c->k_first, c->k_last,
                               c->wide_of, counters, fr1, c->wnodes, c->k_dp);
            uint32_t hc[2];
            EGR_HIP(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, s));
            EGR_HIP(hipStreamSynchronize(s));
            count = hc[1];
            if (hc[0] > nw) c->level_start.push_back(hc[0]);
            nw = hc[0];
            EGR_HIP(hipMemcpyAsync(counters + 1, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, s));
            std::swap(fr0, fr1);
            if (c->level_start.size() > 200) throw EgrCheck{hipErrorInvalidValue, "wide BVH deeper than 200 levels"};
        }
        c->num_wide = nw;
    }
    c->max_depth = (uint32_t)c->level_start.size() - 1;
    refit_boxes(c, s);
    EGR_HIP(hipStreamSynchronize(s));
    c->bvh_valid = true;
}

void egr_bvh_refit(egr_context *c, hipStream_t s, bool fuse_live) {
    const uint32_t n = c->g.count;
    if (!c->bvh_valid || n != c->n_built) throw EgrCheck{hipErrorInvalidValue, "update_bvh: tree was built for a different gaussian count; call rebuild_bvh"};
    c->boxes_are_cubes = c->exact_stats;
    if (n == 0) return;
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)c->pos_of_gid, c->inst_w, c->inst_m, c->aabb, c->exact_stats ? 1 : 0,
                       c->app, fuse_live ? 1 : 0, c->bsph);
    c->live_fresh = fuse_live;
    refit_boxes(c, s);
}

// Host-side structural self check (debug / tests): every child index valid and visited once, internal slot boxes are
// the exact integer union of the child's slots, every leaf slot box contains its gaussian's box (after decoding),
// every gaussian is reachable exactly once.
int egr_bvh_check(egr_context *c, hipStream_t s, std::string &msg) {
    const uint32_t n = c->n_built, nw = c->num_wide;
    if (n == 0) return 0;
    std::vector<uint4> nodes((size_t)nw * EGR_WIDTH);
    std::vector<uint32_t> gop(n);
    std::vector<float> aabb(6 * (size_t)n);
    EGR_HIP(hipStreamSynchronize(s));
    EGR_HIP(hipMemcpy(nodes.data(), c->wnodes, sizeof(uint4) * nodes.size(), hipMemcpyDeviceToHost));
    EGR_HIP(hipMemcpy(gop.data(), c->vals_out, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    EGR_HIP(hipMemcpy(aabb.data(), c->aabb, sizeof(float) * aabb.size(), hipMemcpyDeviceToHost));
    const BvhFrame fr = c->frame;
    const float fo[3] = {fr.ox, fr.oy, fr.oz}, fs[3] = {fr.sx, fr.sy, fr.sz};
    auto unpack = [](uint4 q, uint32_t lo[3], uint32_t hi[3]) {
        lo[0] = q.x & 0xFFFFu, lo[1] = q.x >> 16, lo[2] = q.y & 0xFFFFu, hi[0] = q.y >> 16, hi[1] = q.z & 0xFFFFu, hi[2] = q.z >> 16;
    };
    auto dec = [&](uint32_t q, int a, bool is_lo) -> double {
        if (is_lo && q == 0) return -1e300;
        if (!is_lo && q == 65535) return 1e300;
        return (double)fo[a] + ((double)q - 2.0) / (double)fs[a];
    };
    std::vector<uint8_t> seen(n, 0), wseen(nw, 0);
    char buf[256];
    std::vector<uint32_t> st;
    st.push_back(0);
    wseen[0] = 1;
    while (!st.empty()) {
        const uint32_t w = st.back();
        st.pop_back();
        bool gap = false;
        for (int k = 0; k < EGR_WIDTH; k++) {
            const uint4 q = nodes[(size_t)w * EGR_WIDTH + k];
            uint32_t lo[3], hi[3];
            unpack(q, lo, hi);
            if (q.w == EGR_EMPTY_SLOT) { gap = true; continue; }
            if (gap) { snprintf(buf, sizeof buf, "node %u: slot %d used after an empty slot", w, k); msg = buf; return 10; }
            if (q.w & EGR_LEAF_FLAG) {
                const uint32_t pos = q.w & ~EGR_LEAF_FLAG;
                if (pos >= n || seen[gop[pos]]) { snprintf(buf, sizeof buf, "node %u slot %d: bad/duplicate leaf %u", w, k, pos); msg = buf; return 2; }
                const uint32_t gid = gop[pos];
                seen[gid] = 1;
                const float *b = &aabb[6 * (size_t)gid];
                if (!(b[0] <= b[3])) continue; // unusable gaussian: empty box
                for (int a = 0; a < 3; a++)
                    if (dec(lo[a], a, true) > (double)b[a] || dec(hi[a], a, false) < (double)b[3 + a]) {
                        snprintf(buf, sizeof buf, "node %u slot %d: box does not contain gaussian %u on axis %d", w, k, gid, a); msg = buf; return 4;
                    }
                continue;
            }
            const uint32_t ch = q.w;
            if (ch >= nw || wseen[ch]) { snprintf(buf, sizeof buf, "node %u slot %d: bad/duplicate child %u", w, k, ch); msg = buf; return 6; }
            wseen[ch] = 1;
            uint32_t ulo[3] = {65535u, 65535u, 65535u}, uhi[3] = {0u, 0u, 0u};
            for (int cc = 0; cc < EGR_WIDTH; cc++) {
                uint32_t l[3], h[3];
                unpack(nodes[(size_t)ch * EGR_WIDTH + cc], l, h);
                for (int a = 0; a < 3; a++) ulo[a] = std::min(ulo[a], l[a]), uhi[a] = std::max(uhi[a], h[a]);
            }
            for (int a = 0; a < 3; a++)
                if (lo[a] != ulo[a] || hi[a] != uhi[a]) { snprintf(buf, sizeof buf, "node %u slot %d: box is not the union of child %u", w, k, ch); msg = buf; return 7; }
            st.push_back(ch);
        }
    }
    for (uint32_t i = 0; i < n; i++)
        if (!seen[i]) { snprintf(buf, sizeof buf, "gaussian %u unreachable", i); msg = buf; return 8; }
    for (uint32_t w = 0; w < nw; w++)
        if (!wseen[w]) { snprintf(buf, sizeof buf, "wide node %u unreachable", w); msg = buf; return 9; }
    return 0;
}
