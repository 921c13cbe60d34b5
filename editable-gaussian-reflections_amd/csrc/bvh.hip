// Instance transforms + LBVH build/refit for gfx950. Replaces the reference's K1/K2:
//   _populateBVH / create_transform_matrix   (optix/bvh_wrapper.cu:9-59)
//   OptiX TLAS build / refit                 (optix/bvh_wrapper.h:32-59,118-157; closed source)
//
// Design (MI355X-first, not a translation of OptiX instancing):
//   * per-lane traversal is bound by the texture addresser (one divergent 16-B gather per lane per clock), so a
//     node is ONE 16-byte load: the box is quantised to 16 bits per coordinate in a per-build frame (conservative:
//     one extra cell each side; 0 / 65535 decode to -inf / +inf so boxes that drift out of the frame between
//     rebuilds stay correct), plus a 32-bit link;
//   * leaves are CLUSTERS of 4 Morton-consecutive Gaussians: 4x fewer tree nodes, and the members are tested with
//     the exact object-space cube test directly (their loose world AABBs are never fetched during traversal);
//   * nodes are stored in DFS pre-order with a skip index ("threaded" BVH), so traversal needs NO stack:
//     hit -> node+1, miss -> skip, after a leaf -> node+1;
//   * topology from 63-bit Morton codes of the Gaussian means (Karras 2012), built only on rebuild;
//   * the per-iteration refit is atomics-free and fence-free: internal nodes are bucketed by depth at build
//     time and refitted deepest-first, one launch per depth (kernel boundaries provide the ordering the
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
// record that can never pass the cube test (they are cluster members, so traversal may still look at them).
// ---------------------------------------------------------------------------------------------------------
// Records are stored at the gaussian's MORTON-SORTED position (pos_of_gid), i.e. in leaf order: rays that are close
// in space then read neighbouring 48-B records (same / adjacent cache lines) instead of lines scattered over the
// whole array - the scene's own order is arbitrary (the reference never sorts its clouds).
__global__ void __launch_bounds__(BS) k_instances(uint32_t n, egr_gaussians g, egr_config cfg, const uint32_t *__restrict__ pos_of_gid,
                                                  float4 *__restrict__ inst_w, float4 *__restrict__ inst_m, float *__restrict__ aabb) {
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
        inst_m[3 * slot + a] = make_float4(M0, M1, M2, m[a]);
        float W0 = R[0][a] * inv[a], W1 = R[1][a] * inv[a], W2 = R[2][a] * inv[a];
        Wr[a] = make_float4(W0, W1, W2, -(W0 * m[0] + W1 * m[1] + W2 * m[2]));
        finite = finite && isfinite(W0 + W1 + W2 + Wr[a].w);
        float ext = fabsf(M0) + fabsf(M1) + fabsf(M2);
        // boxes only prune; candidacy is decided by the exact object-space cube test. Pad against fp32 rounding.
        ext = ext * 1.0001f + 4e-7f * (fabsf(m[a]) + ext);
        lo[a] = m[a] - ext;
        hi[a] = m[a] + ext;
        finite = finite && (lo[a] <= hi[a]) && isfinite(lo[a]) && isfinite(hi[a]);
    }
    const bool usable = visible && finite; // NaN / inf parameters can never be hit
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // unusable: object-space origin (2,2,2), zero direction -> outside the unit cube for every ray
        inst_w[3 * slot + a] = usable ? Wr[a] : make_float4(0.f, 0.f, 0.f, 2.f);
        aabb[6 * i + a] = usable ? lo[a] : 3.0e38f;
        aabb[6 * i + 3 + a] = usable ? hi[a] : -3.0e38f;
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
__global__ void __launch_bounds__(BS) k_morton(uint32_t n, const float *__restrict__ mean, BvhFrame fr,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    uint64_t key = 0;
    const float fo[3] = {fr.ox, fr.oy, fr.oz}, fs[3] = {fr.sx, fr.sy, fr.sz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float v = mean[3 * i + a];
        float u = isfinite(v) ? (v - fo[a]) * fs[a] * (1.0f / 65536.0f) : 0.0f; // [0,1) inside the frame
        uint32_t q = (uint32_t)fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
        key |= spread21(q) << (2 - a);
    }
    keys[i] = key;
    vals[i] = i;
}
// clusters: EGR_CLUSTER Morton-consecutive gaussians (sorted positions [C*j, C*j+C)); cluster key = key of its first member
__global__ void __launch_bounds__(BS) k_clusters(uint32_t n, uint32_t nc, const uint64_t *__restrict__ keys_sorted, uint64_t *__restrict__ ckeys) {
    uint32_t j = blockIdx.x * BS + threadIdx.x;
    if (j < nc) ckeys[j] = keys_sorted[EGR_CLUSTER * j];
}
__global__ void __launch_bounds__(BS) k_inverse_perm(uint32_t n, const uint32_t *__restrict__ gid_of_pos, uint32_t *__restrict__ pos_of_gid) {
    uint32_t p = blockIdx.x * BS + threadIdx.x;
    if (p < n) pos_of_gid[gid_of_pos[p]] = p;
}

// ---------------------------------------------------------------------------------------------------------
// Karras 2012 topology over the clusters. Id space: internal i in [0,n-2] -> i, leaf j in [0,n-1] -> (n-1)+j.
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

// Pre-order position of every node: pre = 2*first + (#left turns on the root path); skip = pre + 2*leaves - 1.
__global__ void __launch_bounds__(BS) k_preorder(int n, const int32_t *__restrict__ left, const int32_t *__restrict__ parent,
                                                 const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                                 uint4 *__restrict__ qnodes, uint32_t *__restrict__ leaf_pre,
                                                 uint32_t *__restrict__ node_depth, uint32_t *__restrict__ hist) {
    int id = blockIdx.x * BS + threadIdx.x;
    if (id >= 2 * n - 1) return;
    bool leaf = id >= n - 1;
    uint32_t f = leaf ? (uint32_t)(id - (n - 1)) : first[id];
    uint32_t leaves = leaf ? 1u : last[id] - first[id] + 1u;
    uint32_t turns = 0, depth = 0;
    int x = id;
    while (true) {
        int p = (n == 1) ? -1 : parent[x];
        if (p < 0) break;
        turns += (left[p] == x) ? 1u : 0u;
        depth++;
        x = p;
    }
    uint32_t pre = 2u * f + turns;
    uint32_t skip = pre + 2u * leaves - 1u;
    uint32_t link = leaf ? (EGR_LEAF_FLAG | (uint32_t)(id - (n - 1))) : skip; // leaf payload finalised in k_leaf_boxes
    qnodes[pre] = make_uint4(0xFFFFFFFFu, 0x0000FFFFu, 0u, link); // empty box: lo = 65535, hi = 0
    node_depth[pre] = leaf ? 0xFFFFFFFFu : depth;
    if (leaf) leaf_pre[id - (n - 1)] = pre;
    else atomicAdd(&hist[min(depth, (uint32_t)EGR_MAX_DEPTH_BINS - 1)], 1u);
}
__global__ void __launch_bounds__(BS) k_scatter_depth(uint32_t num_nodes, const uint32_t *__restrict__ node_depth,
                                                      uint32_t *__restrict__ cursor, uint32_t *__restrict__ order) {
    uint32_t p = blockIdx.x * BS + threadIdx.x;
    if (p >= num_nodes) return;
    uint32_t d = node_depth[p];
    if (d == 0xFFFFFFFFu) return;
    uint32_t pos = atomicAdd(&cursor[min(d, (uint32_t)EGR_MAX_DEPTH_BINS - 1)], 1u);
    order[pos] = p;
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
__device__ __forceinline__ uint4 pack_box(const uint32_t lo[3], const uint32_t hi[3], uint32_t link) {
    return make_uint4(lo[0] | (lo[1] << 16), lo[2] | (hi[0] << 16), hi[1] | (hi[2] << 16), link);
}
__device__ __forceinline__ uint4 union_box(uint4 a, uint4 b, uint32_t link) {
    uint32_t lo[3] = {min(a.x & 0xFFFFu, b.x & 0xFFFFu), min(a.x >> 16, b.x >> 16), min(a.y & 0xFFFFu, b.y & 0xFFFFu)};
    uint32_t hi[3] = {max(a.y >> 16, b.y >> 16), max(a.z & 0xFFFFu, b.z & 0xFFFFu), max(a.z >> 16, b.z >> 16)};
    return pack_box(lo, hi, link);
}
__global__ void __launch_bounds__(BS) k_leaf_boxes(uint32_t n, uint32_t nc, const float *__restrict__ aabb, const uint32_t *__restrict__ gid_of_pos,
                                                   const uint32_t *__restrict__ leaf_pre, BvhFrame fr, uint4 *__restrict__ qnodes) {
    uint32_t j = blockIdx.x * BS + threadIdx.x;
    if (j >= nc) return;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int k = 0; k < EGR_CLUSTER; k++) {
        const uint32_t pos = EGR_CLUSTER * j + k;
        if (pos >= n) continue;
        const uint32_t gid = gid_of_pos[pos];
#pragma unroll
        for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], aabb[6 * gid + a]), hi[a] = fmaxf(hi[a], aabb[6 * gid + 3 + a]);
    }
    const uint32_t p = leaf_pre[j];
    const uint32_t link = EGR_LEAF_FLAG | j; // members = sorted positions [C*j, C*j+C)
    const float fo[3] = {fr.ox, fr.oy, fr.oz}, fs[3] = {fr.sx, fr.sy, fr.sz};
    uint32_t ql[3], qh[3];
    const bool empty = !(lo[0] <= hi[0]);
#pragma unroll
    for (int a = 0; a < 3; a++) ql[a] = empty ? 65535u : quant_lo(lo[a], fo[a], fs[a]), qh[a] = empty ? 0u : quant_hi(hi[a], fo[a], fs[a]);
    qnodes[p] = pack_box(ql, qh, link);
}
// One depth level: box(p) = box(left = p+1) U box(right). right = skip(left) for an internal left child, left+1 for a
// leaf. Children are one level deeper and were written by an earlier launch.
__device__ __forceinline__ void refit_one(uint32_t p, uint4 *nodes) {
    const uint32_t l = p + 1;
    const uint4 ln = nodes[l];
    const uint32_t r = (ln.w & EGR_LEAF_FLAG) ? l + 1 : ln.w;
    const uint4 rn = nodes[r];
    nodes[p] = union_box(ln, rn, nodes[p].w);
}
__global__ void __launch_bounds__(BS) k_refit_level(uint32_t count, const uint32_t *__restrict__ order, uint4 *__restrict__ nodes) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= count) return;
    refit_one(order[i], nodes);
}
// The shallow levels hold few nodes each; refit all of them in one single-workgroup launch instead of one
// launch per level (levels processed deepest first, separated by workgroup barriers; same CU -> same L1/L2).
__global__ void __launch_bounds__(1024) k_refit_top(int top_levels, const uint32_t *__restrict__ level_start,
                                                    const uint32_t *__restrict__ order, uint4 *nodes) {
    for (int d = top_levels - 1; d >= 0; d--) {
        uint32_t b = level_start[d], e = level_start[d + 1];
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) refit_one(order[i], nodes);
        __threadfence_block();
        __syncthreads();
    }
}

template <class T> void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}
template <class T> void dalloc(T *&p, size_t count) {
    dfree(p);
    EGR_HIP(hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)));
}

} // namespace

void egr_bvh_free(egr_context *c) {
    dfree(c->qnodes), dfree(c->pos_of_gid), dfree(c->inst_w), dfree(c->inst_m), dfree(c->app), dfree(c->aabb), dfree(c->leaf_pre);
    dfree(c->depth_order), dfree(c->sort_tmp), dfree(c->keys_in), dfree(c->keys_out), dfree(c->vals_in);
    dfree(c->vals_out), dfree(c->k_left), dfree(c->k_right), dfree(c->k_parent), dfree(c->k_first), dfree(c->k_last);
    dfree(c->node_depth), dfree(c->scratch_u32);
    c->n_alloc = 0;
    c->n_built = 0;
    c->bvh_valid = false;
}

void egr_bvh_reserve(egr_context *c, uint32_t n) {
    if (n <= c->n_alloc && c->qnodes) return;
    uint32_t cap = std::max<uint32_t>(n + n / 8, 256); // head-room: the reference grows by +75k far-field points
    uint32_t ccap = cap / EGR_CLUSTER + 64;             // clusters
    dalloc(c->qnodes, 2 * (size_t)ccap + 16);
    dalloc(c->pos_of_gid, cap);
    dalloc(c->inst_w, 3 * (size_t)cap);
    dalloc(c->inst_m, 3 * (size_t)cap);
    dalloc(c->app, 3 * (size_t)cap);
    dalloc(c->aabb, 6 * (size_t)cap);
    dalloc(c->leaf_pre, ccap);
    dalloc(c->depth_order, ccap);
    dalloc(c->keys_in, cap), dalloc(c->keys_out, cap), dalloc(c->vals_in, cap), dalloc(c->vals_out, cap);
    dalloc(c->k_left, ccap), dalloc(c->k_right, ccap), dalloc(c->k_parent, 2 * (size_t)ccap);
    dalloc(c->k_first, ccap), dalloc(c->k_last, ccap);
    dalloc(c->node_depth, 2 * (size_t)ccap);
    if (!c->scratch_u32) dalloc(c->scratch_u32, 16 + 2 * EGR_MAX_DEPTH_BINS + 8);
    size_t bytes = 0;
    EGR_HIP(rocprim::radix_sort_pairs(nullptr, bytes, c->keys_in, c->keys_out, c->vals_in, c->vals_out, (size_t)cap, 0, 63, 0));
    dfree(c->sort_tmp);
    EGR_HIP(hipMalloc(&c->sort_tmp, bytes));
    c->sort_tmp_bytes = bytes;
    c->n_alloc = cap;
    c->bvh_valid = false;
}

static void refit_boxes(egr_context *c, hipStream_t s) {
    const uint32_t nc = c->n_clusters;
    hipLaunchKernelGGL(k_leaf_boxes, dim3(nblk(nc)), dim3(BS), 0, s, c->n_built, nc, c->aabb, c->vals_out, c->leaf_pre, c->frame, c->qnodes);
    // depth_start[d]..depth_start[d+1] = internal nodes at depth d. Deep levels: one launch each.
    // Shallow levels (cumulatively <= 8192 nodes): one single-workgroup launch.
    int top = 0;
    while (top <= (int)c->max_depth && c->depth_start[top + 1] <= 8192u) top++;
    for (int d = (int)c->max_depth; d >= top; d--) {
        uint32_t b = c->depth_start[d], e = c->depth_start[d + 1];
        if (e > b) hipLaunchKernelGGL(k_refit_level, dim3(nblk(e - b)), dim3(BS), 0, s, e - b, c->depth_order + b, c->qnodes);
    }
    if (top > 0 && c->depth_start[top] > 0)
        hipLaunchKernelGGL(k_refit_top, dim3(1), dim3(1024), 0, s, top, c->scratch_u32 + 16 + EGR_MAX_DEPTH_BINS, c->depth_order,
                           c->qnodes);
}

void egr_bvh_rebuild(egr_context *c, hipStream_t s) {
    const uint32_t n = c->g.count;
    egr_bvh_reserve(c, n);
    c->n_built = n;
    c->n_clusters = (n + EGR_CLUSTER - 1) / EGR_CLUSTER;
    c->max_depth = 0;
    c->depth_start.assign(EGR_MAX_DEPTH_BINS + 1, 0);
    c->frame = BvhFrame{0.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    if (n == 0) {
        c->bvh_valid = true;
        return;
    }
    const uint32_t nc = c->n_clusters;
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)nullptr, c->inst_w, c->inst_m, c->aabb); // boxes for frame + leaves
    uint32_t *bounds = c->scratch_u32, *hist = c->scratch_u32 + 16, *cursor = c->scratch_u32 + 16 + EGR_MAX_DEPTH_BINS;
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
    uint64_t *ckeys = c->keys_in; // free after the sort
    hipLaunchKernelGGL(k_clusters, dim3(nblk(nc)), dim3(BS), 0, s, n, nc, c->keys_out, ckeys);
    hipLaunchKernelGGL(k_inverse_perm, dim3(nblk(n)), dim3(BS), 0, s, n, c->vals_out, c->pos_of_gid);
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)c->pos_of_gid, c->inst_w, c->inst_m, c->aabb); // records in leaf order
    if (nc >= 2)
        hipLaunchKernelGGL(k_karras, dim3(nblk(nc - 1)), dim3(BS), 0, s, (int)nc, ckeys, c->k_left, c->k_right, c->k_parent, c->k_first,
                           c->k_last);
    EGR_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 2 * EGR_MAX_DEPTH_BINS + 32, s));
    hipLaunchKernelGGL(k_preorder, dim3(nblk(2 * (uint64_t)nc - 1)), dim3(BS), 0, s, (int)nc, c->k_left, c->k_parent, c->k_first,
                       c->k_last, c->qnodes, c->leaf_pre, c->node_depth, hist);
    std::vector<uint32_t> h(EGR_MAX_DEPTH_BINS);
    EGR_HIP(hipMemcpyAsync(h.data(), hist, sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyDeviceToHost, s));
    EGR_HIP(hipStreamSynchronize(s));
    if (h[EGR_MAX_DEPTH_BINS - 1] != 0) throw EgrCheck{hipErrorInvalidValue, "LBVH deeper than EGR_MAX_DEPTH_BINS"};
    uint32_t maxd = 0;
    for (int d = 0; d < EGR_MAX_DEPTH_BINS; d++) {
        c->depth_start[d + 1] = c->depth_start[d] + h[d];
        if (h[d]) maxd = d;
    }
    c->max_depth = maxd;
    // cursors double as the device copy of level_start for k_refit_top (it reads entries [0, top])
    EGR_HIP(hipMemcpyAsync(cursor, c->depth_start.data(), sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyHostToDevice, s));
    // scatter needs its own running cursors: a second copy placed in vals_in (free after the sort)
    uint32_t *run = c->vals_in;
    EGR_HIP(hipMemcpyAsync(run, c->depth_start.data(), sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scatter_depth, dim3(nblk(2 * (uint64_t)nc - 1)), dim3(BS), 0, s, 2 * nc - 1, c->node_depth, run, c->depth_order);
    refit_boxes(c, s);
    EGR_HIP(hipStreamSynchronize(s));
    c->bvh_valid = true;
}

void egr_bvh_refit(egr_context *c, hipStream_t s) {
    const uint32_t n = c->g.count;
    if (!c->bvh_valid || n != c->n_built) throw EgrCheck{hipErrorInvalidValue, "update_bvh: tree was built for a different gaussian count; call rebuild_bvh"};
    if (n == 0) return;
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, (const uint32_t *)c->pos_of_gid, c->inst_w, c->inst_m, c->aabb);
    refit_boxes(c, s);
}

// Host-side structural self check (debug / tests): pre-order links, exact integer unions, every leaf box contains
// its members' boxes (after decoding), every gaussian is a member of exactly one reachable cluster.
int egr_bvh_check(egr_context *c, hipStream_t s, std::string &msg) {
    const uint32_t n = c->n_built, nc = c->n_clusters;
    if (n == 0) return 0;
    const uint32_t nn = 2 * nc - 1;
    std::vector<uint4> nodes(nn);
    std::vector<uint32_t> gop(n);
    std::vector<float> aabb(6 * (size_t)n);
    EGR_HIP(hipStreamSynchronize(s));
    EGR_HIP(hipMemcpy(nodes.data(), c->qnodes, sizeof(uint4) * nn, hipMemcpyDeviceToHost));
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
    std::vector<uint8_t> seen(n, 0), cseen(nc, 0);
    char buf[256];
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({0, nn});
    while (!st.empty()) {
        auto [p, end] = st.back();
        st.pop_back();
        const uint4 q = nodes[p];
        uint32_t lo[3], hi[3];
        unpack(q, lo, hi);
        if (q.w & EGR_LEAF_FLAG) {
            const uint32_t j = q.w & ~EGR_LEAF_FLAG;
            if (end != p + 1) { snprintf(buf, sizeof buf, "leaf %u does not end its subtree (%u)", p, end); msg = buf; return 3; }
            if (j >= nc || cseen[j]) { snprintf(buf, sizeof buf, "leaf %u: bad/duplicate cluster %u", p, j); msg = buf; return 2; }
            cseen[j] = 1;
            uint32_t id[EGR_CLUSTER];
            for (int k = 0; k < EGR_CLUSTER; k++) id[k] = (EGR_CLUSTER * j + k < n) ? gop[EGR_CLUSTER * j + k] : 0xFFFFFFFFu;
            for (int k = 0; k < EGR_CLUSTER; k++) {
                if (id[k] == 0xFFFFFFFFu) continue;
                if (id[k] >= n || seen[id[k]]) { snprintf(buf, sizeof buf, "cluster %u: bad/duplicate member %u", j, id[k]); msg = buf; return 9; }
                seen[id[k]] = 1;
                const float *b = &aabb[6 * (size_t)id[k]];
                if (!(b[0] <= b[3])) continue; // unusable member has an empty box
                for (int a = 0; a < 3; a++)
                    if (dec(lo[a], a, true) > (double)b[a] || dec(hi[a], a, false) < (double)b[3 + a]) {
                        snprintf(buf, sizeof buf, "leaf %u (cluster %u) box does not contain member %u on axis %d", p, j, id[k], a); msg = buf; return 4;
                    }
            }
            continue;
        }
        if (q.w != end) { snprintf(buf, sizeof buf, "node %u: skip %u != subtree end %u", p, q.w, end); msg = buf; return 1; }
        const uint32_t l = p + 1;
        if (l >= nn) { msg = "internal node without child"; return 5; }
        const uint4 ln = nodes[l];
        const uint32_t r = (ln.w & EGR_LEAF_FLAG) ? l + 1 : ln.w;
        if (r <= l || r >= end) { snprintf(buf, sizeof buf, "node %u: right child %u outside (%u,%u)", p, r, l, end); msg = buf; return 6; }
        const uint4 rn = nodes[r];
        uint32_t llo[3], lhi[3], rlo[3], rhi[3];
        unpack(ln, llo, lhi), unpack(rn, rlo, rhi);
        for (int a = 0; a < 3; a++)
            if (lo[a] != std::min(llo[a], rlo[a]) || hi[a] != std::max(lhi[a], rhi[a])) {
                snprintf(buf, sizeof buf, "node %u: box is not the union of children %u,%u", p, l, r); msg = buf; return 7;
            }
        st.push_back({l, r});
        st.push_back({r, end});
    }
    for (uint32_t i = 0; i < n; i++)
        if (!seen[i]) { snprintf(buf, sizeof buf, "gaussian %u unreachable", i); msg = buf; return 8; }
    return 0;
}
