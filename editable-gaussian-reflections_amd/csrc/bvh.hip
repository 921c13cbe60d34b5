// Instance transforms + LBVH build/refit for gfx950. Replaces the reference's K1/K2:
//   _populateBVH / create_transform_matrix   (optix/bvh_wrapper.cu:9-59)
//   OptiX TLAS build / refit                 (optix/bvh_wrapper.h:32-59,118-157; closed source)
//
// Design (MI355X-first, not a translation of OptiX instancing):
//   * one 32-byte node = two dwordx4 loads; nodes are stored in DFS pre-order with a skip index
//     ("threaded" BVH), so traversal needs NO stack: hit -> node+1, miss -> skip. Descending to the left
//     child is a sequential read of the same 128-B line.
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
// mask (bvh_wrapper.cu:55): sigma > 0 && any(size > 0); masked-out instances get an empty box (lo > hi).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BS) k_instances(uint32_t n, egr_gaussians g, egr_config cfg, float4 *__restrict__ inst_w,
                                                  float4 *__restrict__ inst_m, float *__restrict__ aabb,
                                                  float4 *__restrict__ nodes, const uint32_t *__restrict__ leaf_pre) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
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
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float M0 = s[0] * R[a][0], M1 = s[1] * R[a][1], M2 = s[2] * R[a][2];
        inst_m[3 * i + a] = make_float4(M0, M1, M2, m[a]);
        float W0 = R[0][a] * inv[a], W1 = R[1][a] * inv[a], W2 = R[2][a] * inv[a];
        inst_w[3 * i + a] = make_float4(W0, W1, W2, -(W0 * m[0] + W1 * m[1] + W2 * m[2]));
        float ext = fabsf(M0) + fabsf(M1) + fabsf(M2);
        // boxes only prune; candidacy is decided by the exact object-space cube test. Pad against fp32 rounding.
        ext = ext * 1.0001f + 4e-7f * (fabsf(m[a]) + ext);
        lo[a] = visible ? m[a] - ext : 3.0e38f;
        hi[a] = visible ? m[a] + ext : -3.0e38f;
        if (!(lo[a] <= hi[a]) && visible) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } // NaN parameters -> never hit
    }
#pragma unroll
    for (int a = 0; a < 3; a++) aabb[6 * i + a] = lo[a], aabb[6 * i + 3 + a] = hi[a];
    if (leaf_pre) { // refit path: the leaf's slot in the pre-order array is known
        uint32_t p = leaf_pre[i];
        float4 a0 = nodes[2 * p], a1 = nodes[2 * p + 1];
        nodes[2 * p] = make_float4(lo[0], lo[1], lo[2], a0.w);
        nodes[2 * p + 1] = make_float4(hi[0], hi[1], hi[2], a1.w);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Morton codes of the means
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float f_unordered(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__global__ void k_bounds_init(uint32_t *b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(BS) k_bounds(uint32_t n, const float *__restrict__ mean, uint32_t *__restrict__ b) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    float v[3] = {0, 0, 0};
    bool ok = false;
    if (i < n) {
        v[0] = mean[3 * i], v[1] = mean[3 * i + 1], v[2] = mean[3 * i + 2];
        ok = isfinite(v[0] + v[1] + v[2]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        uint32_t lo = ok ? f_ordered(v[a]) : 0xFFFFFFFFu, hi = ok ? f_ordered(v[a]) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&b[a], lo);
            atomicMax(&b[3 + a], hi);
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
__global__ void __launch_bounds__(BS) k_morton(uint32_t n, const float *__restrict__ mean, const uint32_t *__restrict__ b,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    uint64_t key = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float lo = f_unordered(b[a]), hi = f_unordered(b[3 + a]);
        float v = mean[3 * i + a];
        float ext = hi - lo;
        float u = (ext > 0.0f && isfinite(v)) ? (v - lo) / ext : 0.0f;
        uint32_t q = (uint32_t)fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
        key |= spread21(q) << (2 - a);
    }
    keys[i] = key;
    vals[i] = i;
}

// ---------------------------------------------------------------------------------------------------------
// Karras 2012 topology. Id space: internal i in [0,n-2] -> i, leaf j in [0,n-1] -> (n-1)+j.
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
                                                 const uint32_t *__restrict__ sorted_ids, float4 *__restrict__ nodes,
                                                 uint32_t *__restrict__ leaf_pre, uint32_t *__restrict__ node_depth,
                                                 uint32_t *__restrict__ hist) {
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
    uint32_t prim = leaf ? sorted_ids[id - (n - 1)] : EGR_INTERNAL_NODE;
    nodes[2 * pre] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, u2f(skip));
    nodes[2 * pre + 1] = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, u2f(prim));
    node_depth[pre] = leaf ? 0xFFFFFFFFu : depth;
    if (leaf) leaf_pre[prim] = pre;
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
__global__ void __launch_bounds__(BS) k_leaf_boxes(uint32_t n, const float *__restrict__ aabb, const uint32_t *__restrict__ leaf_pre,
                                                   float4 *__restrict__ nodes) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= n) return;
    uint32_t p = leaf_pre[i];
    float4 a0 = nodes[2 * p], a1 = nodes[2 * p + 1];
    nodes[2 * p] = make_float4(aabb[6 * i], aabb[6 * i + 1], aabb[6 * i + 2], a0.w);
    nodes[2 * p + 1] = make_float4(aabb[6 * i + 3], aabb[6 * i + 4], aabb[6 * i + 5], a1.w);
}
// One depth level: box(p) = box(left = p+1) U box(right = skip(left)). Children are one level deeper and were
// written by an earlier launch.
__global__ void __launch_bounds__(BS) k_refit_level(uint32_t count, const uint32_t *__restrict__ order, float4 *__restrict__ nodes) {
    uint32_t i = blockIdx.x * BS + threadIdx.x;
    if (i >= count) return;
    uint32_t p = order[i];
    uint32_t l = p + 1;
    float4 l0 = nodes[2 * l], l1 = nodes[2 * l + 1];
    uint32_t r = f2u(l0.w);
    float4 r0 = nodes[2 * r], r1 = nodes[2 * r + 1];
    float4 p0 = nodes[2 * p], p1 = nodes[2 * p + 1];
    nodes[2 * p] = make_float4(fminf(l0.x, r0.x), fminf(l0.y, r0.y), fminf(l0.z, r0.z), p0.w);
    nodes[2 * p + 1] = make_float4(fmaxf(l1.x, r1.x), fmaxf(l1.y, r1.y), fmaxf(l1.z, r1.z), p1.w);
}
// The shallow levels hold few nodes each; refit all of them in one single-workgroup launch instead of one
// launch per level (levels processed deepest first, separated by workgroup barriers; same CU -> same L1/L2).
__global__ void __launch_bounds__(1024) k_refit_top(int top_levels, const uint32_t *__restrict__ level_start,
                                                    const uint32_t *__restrict__ order, float4 *nodes) {
    for (int d = top_levels - 1; d >= 0; d--) {
        uint32_t b = level_start[d], e = level_start[d + 1];
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
            uint32_t p = order[i];
            uint32_t l = p + 1;
            float4 l0 = nodes[2 * l], l1 = nodes[2 * l + 1];
            uint32_t r = f2u(l0.w);
            float4 r0 = nodes[2 * r], r1 = nodes[2 * r + 1];
            float4 p0 = nodes[2 * p], p1 = nodes[2 * p + 1];
            nodes[2 * p] = make_float4(fminf(l0.x, r0.x), fminf(l0.y, r0.y), fminf(l0.z, r0.z), p0.w);
            nodes[2 * p + 1] = make_float4(fmaxf(l1.x, r1.x), fmaxf(l1.y, r1.y), fmaxf(l1.z, r1.z), p1.w);
        }
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
    dfree(c->nodes), dfree(c->inst_w), dfree(c->inst_m), dfree(c->app), dfree(c->aabb), dfree(c->leaf_pre);
    dfree(c->depth_order), dfree(c->sort_tmp), dfree(c->keys_in), dfree(c->keys_out), dfree(c->vals_in);
    dfree(c->vals_out), dfree(c->k_left), dfree(c->k_right), dfree(c->k_parent), dfree(c->k_first), dfree(c->k_last);
    dfree(c->node_depth), dfree(c->scratch_u32);
    c->n_alloc = 0;
    c->n_built = 0;
    c->bvh_valid = false;
}

void egr_bvh_reserve(egr_context *c, uint32_t n) {
    if (n <= c->n_alloc && c->nodes) return;
    uint32_t cap = std::max<uint32_t>(n + n / 8, 256); // head-room: the reference grows by +75k far-field points
    dalloc(c->nodes, 2 * (2 * (size_t)cap));
    dalloc(c->inst_w, 3 * (size_t)cap);
    dalloc(c->inst_m, 3 * (size_t)cap);
    dalloc(c->app, 3 * (size_t)cap);
    dalloc(c->aabb, 6 * (size_t)cap);
    dalloc(c->leaf_pre, cap);
    dalloc(c->depth_order, cap);
    dalloc(c->keys_in, cap), dalloc(c->keys_out, cap), dalloc(c->vals_in, cap), dalloc(c->vals_out, cap);
    dalloc(c->k_left, cap), dalloc(c->k_right, cap), dalloc(c->k_parent, 2 * (size_t)cap);
    dalloc(c->k_first, cap), dalloc(c->k_last, cap);
    dalloc(c->node_depth, 2 * (size_t)cap);
    if (!c->scratch_u32) dalloc(c->scratch_u32, 16 + 2 * EGR_MAX_DEPTH_BINS + 8);
    size_t bytes = 0;
    EGR_HIP(rocprim::radix_sort_pairs(nullptr, bytes, c->keys_in, c->keys_out, c->vals_in, c->vals_out, (size_t)cap, 0, 63, 0));
    dfree(c->sort_tmp);
    EGR_HIP(hipMalloc(&c->sort_tmp, bytes));
    c->sort_tmp_bytes = bytes;
    c->n_alloc = cap;
    c->bvh_valid = false;
}

static void refit_levels(egr_context *c, hipStream_t s) {
    // depth_start[d]..depth_start[d+1] = internal nodes at depth d. Deep levels: one launch each.
    // Shallow levels (cumulatively <= 8192 nodes): one single-workgroup launch.
    int top = 0;
    while (top <= (int)c->max_depth && c->depth_start[top + 1] <= 8192u) top++;
    for (int d = (int)c->max_depth; d >= top; d--) {
        uint32_t b = c->depth_start[d], e = c->depth_start[d + 1];
        if (e > b) hipLaunchKernelGGL(k_refit_level, dim3(nblk(e - b)), dim3(BS), 0, s, e - b, c->depth_order + b, c->nodes);
    }
    if (top > 0)
        hipLaunchKernelGGL(k_refit_top, dim3(1), dim3(1024), 0, s, top, c->scratch_u32 + 16 + EGR_MAX_DEPTH_BINS, c->depth_order,
                           c->nodes);
}

void egr_bvh_rebuild(egr_context *c, hipStream_t s) {
    const uint32_t n = c->g.count;
    egr_bvh_reserve(c, n);
    c->n_built = n;
    c->max_depth = 0;
    c->depth_start.assign(2, 0);
    if (n == 0) {
        c->bvh_valid = true;
        return;
    }
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, c->inst_w, c->inst_m, c->aabb, c->nodes,
                       (const uint32_t *)nullptr);
    uint32_t *bounds = c->scratch_u32, *hist = c->scratch_u32 + 16, *cursor = c->scratch_u32 + 16 + EGR_MAX_DEPTH_BINS;
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, s, bounds);
    hipLaunchKernelGGL(k_bounds, dim3(nblk(n)), dim3(BS), 0, s, n, c->g.mean, bounds);
    hipLaunchKernelGGL(k_morton, dim3(nblk(n)), dim3(BS), 0, s, n, c->g.mean, bounds, c->keys_in, c->vals_in);
    size_t bytes = c->sort_tmp_bytes;
    EGR_HIP(rocprim::radix_sort_pairs(c->sort_tmp, bytes, c->keys_in, c->keys_out, c->vals_in, c->vals_out, (size_t)n, 0, 63, s));
    if (n >= 2)
        hipLaunchKernelGGL(k_karras, dim3(nblk(n - 1)), dim3(BS), 0, s, (int)n, c->keys_out, c->k_left, c->k_right, c->k_parent,
                           c->k_first, c->k_last);
    EGR_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * 2 * EGR_MAX_DEPTH_BINS + 32, s));
    hipLaunchKernelGGL(k_preorder, dim3(nblk(2 * (uint64_t)n - 1)), dim3(BS), 0, s, (int)n, c->k_left, c->k_parent, c->k_first,
                       c->k_last, c->vals_out, c->nodes, c->leaf_pre, c->node_depth, hist);
    std::vector<uint32_t> h(EGR_MAX_DEPTH_BINS);
    EGR_HIP(hipMemcpyAsync(h.data(), hist, sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyDeviceToHost, s));
    EGR_HIP(hipStreamSynchronize(s));
    if (h[EGR_MAX_DEPTH_BINS - 1] != 0) throw EgrCheck{hipErrorInvalidValue, "LBVH deeper than EGR_MAX_DEPTH_BINS"};
    c->depth_start.assign(EGR_MAX_DEPTH_BINS + 1, 0);
    uint32_t maxd = 0;
    for (int d = 0; d < EGR_MAX_DEPTH_BINS; d++) {
        c->depth_start[d + 1] = c->depth_start[d] + h[d];
        if (h[d]) maxd = d;
    }
    c->max_depth = maxd;
    // cursors double as the device copy of level_start for k_refit_top (it reads entries [0, top])
    EGR_HIP(hipMemcpyAsync(cursor, c->depth_start.data(), sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyHostToDevice, s));
    // scatter needs its own running cursors: use a second copy placed in keys_in (free after the sort)
    uint32_t *run = reinterpret_cast<uint32_t *>(c->keys_in);
    EGR_HIP(hipMemcpyAsync(run, c->depth_start.data(), sizeof(uint32_t) * EGR_MAX_DEPTH_BINS, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_scatter_depth, dim3(nblk(2 * (uint64_t)n - 1)), dim3(BS), 0, s, 2 * n - 1, c->node_depth, run, c->depth_order);
    hipLaunchKernelGGL(k_leaf_boxes, dim3(nblk(n)), dim3(BS), 0, s, n, c->aabb, c->leaf_pre, c->nodes);
    refit_levels(c, s);
    EGR_HIP(hipStreamSynchronize(s));
    c->bvh_valid = true;
}

void egr_bvh_refit(egr_context *c, hipStream_t s) {
    const uint32_t n = c->g.count;
    if (!c->bvh_valid || n != c->n_built) throw EgrCheck{hipErrorInvalidValue, "update_bvh: tree was built for a different gaussian count; call rebuild_bvh"};
    if (n == 0) return;
    hipLaunchKernelGGL(k_instances, dim3(nblk(n)), dim3(BS), 0, s, n, c->g, c->cfg, c->inst_w, c->inst_m, c->aabb, c->nodes,
                       (const uint32_t *)c->leaf_pre);
    refit_levels(c, s);
}

// Host-side structural self check (debug / tests).
int egr_bvh_check(egr_context *c, hipStream_t s, std::string &msg) {
    const uint32_t n = c->n_built;
    if (n == 0) return 0;
    const uint32_t nn = 2 * n - 1;
    std::vector<float4> nodes(2 * (size_t)nn);
    std::vector<float> aabb(6 * (size_t)n);
    EGR_HIP(hipStreamSynchronize(s));
    EGR_HIP(hipMemcpy(nodes.data(), c->nodes, sizeof(float4) * nodes.size(), hipMemcpyDeviceToHost));
    EGR_HIP(hipMemcpy(aabb.data(), c->aabb, sizeof(float) * aabb.size(), hipMemcpyDeviceToHost));
    auto U = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    std::vector<uint8_t> seen(n, 0);
    char buf[256];
    // recursive structure via explicit stack of (node, end)
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({0, nn});
    while (!st.empty()) {
        auto [p, end] = st.back();
        st.pop_back();
        uint32_t skip = U(nodes[2 * p].w), prim = U(nodes[2 * p + 1].w);
        if (skip != end) { snprintf(buf, sizeof buf, "node %u: skip %u != subtree end %u", p, skip, end); msg = buf; return 1; }
        if (prim != EGR_INTERNAL_NODE) {
            if (prim >= n || seen[prim]) { snprintf(buf, sizeof buf, "leaf %u: bad/duplicate prim %u", p, prim); msg = buf; return 2; }
            seen[prim] = 1;
            if (skip != p + 1) { msg = "leaf skip != p+1"; return 3; }
            const float *b = &aabb[6 * prim];
            if (nodes[2 * p].x != b[0] || nodes[2 * p].y != b[1] || nodes[2 * p].z != b[2] || nodes[2 * p + 1].x != b[3] ||
                nodes[2 * p + 1].y != b[4] || nodes[2 * p + 1].z != b[5]) { snprintf(buf, sizeof buf, "leaf %u box != instance %u box", p, prim); msg = buf; return 4; }
            continue;
        }
        uint32_t l = p + 1;
        if (l >= nn) { msg = "internal node without child"; return 5; }
        uint32_t r = U(nodes[2 * l].w);
        if (r <= l || r >= end) { snprintf(buf, sizeof buf, "node %u: right child %u outside (%u,%u)", p, r, l, end); msg = buf; return 6; }
        float4 l0 = nodes[2 * l], l1 = nodes[2 * l + 1], r0 = nodes[2 * r], r1 = nodes[2 * r + 1], p0 = nodes[2 * p], p1 = nodes[2 * p + 1];
        if (p0.x != std::min(l0.x, r0.x) || p0.y != std::min(l0.y, r0.y) || p0.z != std::min(l0.z, r0.z) || p1.x != std::max(l1.x, r1.x) ||
            p1.y != std::max(l1.y, r1.y) || p1.z != std::max(l1.z, r1.z)) { snprintf(buf, sizeof buf, "node %u: box is not the union of children %u,%u", p, l, r); msg = buf; return 7; }
        st.push_back({l, r});
        st.push_back({r, end});
    }
    for (uint32_t i = 0; i < n; i++)
        if (!seen[i]) { snprintf(buf, sizeof buf, "gaussian %u unreachable", i); msg = buf; return 8; }
    return 0;
}
