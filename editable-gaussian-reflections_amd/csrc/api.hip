// extern "C" surface of libegr_hip.so (see include/egr_raytracer.h). Host-only code; the kernels live in
// bvh.hip and trace.hip. The product fails loudly: every HIP error becomes a non-zero return + message.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "egr_internal.hpp"
#include "egr_device.hpp" // egr_div_rn / egr_sqrt_rn: the debug kernel below runs exactly the functions the hot kernels use

void egr_copy_final_to_denoised(egr_context *c, hipStream_t s);
void egr_denoise_atrous(egr_context *c, hipStream_t s); // denoise.hip
void egr_export_step_hits(egr_context *c, int32_t *host_out, hipStream_t s); // trace.hip
void egr_export_hit_hash(egr_context *c, uint64_t *host_out, hipStream_t s); // trace.hip
void egr_upload_targets(egr_context *c, const float *const chw[6], hipStream_t s); // trace.hip
void egr_set_camera_launch(egr_context *c, const float *R, const float *centre, float fov, float znear, float zfar, hipStream_t s); // trace.hip

void egr_stamp_begin(egr_context *c, const char *name, hipStream_t s) {
    if (!c->timing) return;
    if (c->stamps_used == c->stamps.size()) {
        KernelStamp k{name, nullptr, nullptr};
        EGR_HIP(hipEventCreate(&k.start));
        EGR_HIP(hipEventCreate(&k.stop));
        c->stamps.push_back(k);
    }
    c->stamps[c->stamps_used].name = name;
    EGR_HIP(hipEventRecord(c->stamps[c->stamps_used].start, s));
}
void egr_stamp_end(egr_context *c, hipStream_t s) {
    if (!c->timing) return;
    EGR_HIP(hipEventRecord(c->stamps[c->stamps_used].stop, s));
    c->stamps_used++;
}

namespace {
template <class F> int guarded(egr_context *c, F &&f) {
    try {
        int prev = 0;
        EGR_HIP(hipGetDevice(&prev));
        if (prev != c->device) EGR_HIP(hipSetDevice(c->device));
        f();
        if (prev != c->device) EGR_HIP(hipSetDevice(prev));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) throw EgrCheck{e, "kernel launch"};
        return 0;
    } catch (const EgrCheck &e) {
        char buf[512];
        snprintf(buf, sizeof buf, "libegr_hip: %s failed: %s", e.what, hipGetErrorString(e.e));
        c->last_error = buf;
        return 1;
    } catch (const std::exception &e) {
        c->last_error = std::string("libegr_hip: ") + e.what();
        return 1;
    }
}
} // namespace

extern "C" {

const char *egr_version(void) { return "egr-hip 0.6 (gfx950)"; } // 0.6: egr_grad_delta_consumed (round 5); 0.4: egr_counters grew (round 3), egr_get_counters_ex, egr_set_rays_per_task; 0.5: egr_set_team_help

int egr_create(egr_context **out, int device, int width, int height, int64_t ppll_forward_size, int64_t ppll_backward_size) {
    if (!out || width <= 0 || height <= 0) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "libegr_hip: no HIP device available (this library has no CPU fallback)\n");
        return 2;
    }
    egr_context *c = new egr_context();
    c->device = device, c->width = width, c->height = height;
    c->fwd_capacity = ppll_forward_size > 0 ? ppll_forward_size : 1, c->bwd_capacity = ppll_backward_size > 0 ? ppll_backward_size : 1;
    if (const char *e = getenv("EGR_DENOISE")) c->denoise_mode = atoi(e);
    if (const char *e = getenv("EGR_TEAM_HELP")) c->team_help = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("EGR_STRANDS")) c->strands = std::max(1, std::min(EGR_MAX_STRANDS, atoi(e)));
    int rc = guarded(c, [&] {
        egr_trace_alloc(c);
        EGR_HIP(hipEventCreate(&c->ev_rt0)), EGR_HIP(hipEventCreate(&c->ev_rt1));
        EGR_HIP(hipEventCreate(&c->ev_ub0)), EGR_HIP(hipEventCreate(&c->ev_ub1));
    });
    if (rc) {
        fprintf(stderr, "%s\n", c->last_error.c_str());
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

void egr_destroy(egr_context *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    egr_trace_free(c);
    egr_bvh_free(c);
    for (auto &k : c->stamps) (void)hipEventDestroy(k.start), (void)hipEventDestroy(k.stop);
    if (c->ev_rt0) (void)hipEventDestroy(c->ev_rt0), (void)hipEventDestroy(c->ev_rt1), (void)hipEventDestroy(c->ev_ub0), (void)hipEventDestroy(c->ev_ub1);
    delete c;
}

int egr_bind(egr_context *c, const egr_camera *camera, const egr_config *config, const egr_framebuffer *framebuffer,
             const egr_metadata *metadata, const egr_stats *stats) {
    if (!c || !camera || !config || !framebuffer || !metadata || !stats) return 1;
    c->cam = *camera, c->cfg = *config, c->fb = *framebuffer, c->meta = *metadata, c->stats = *stats;
    c->bound = true;
    c->live_fresh = false; // other config scalars: the live records of a fused update are not trusted across a re-bind
    return 0;
}

int egr_set_gaussians(egr_context *c, const egr_gaussians *g) {
    if (!c || !g) return 1;
    c->g = *g;
    c->have_gaussians = true;
    c->live_fresh = false;
    return guarded(c, [&] { egr_bvh_reserve(c, g->count); });
}

int egr_set_partition(egr_context *c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return 1;
    c->rank = rank, c->world = world;
    return guarded(c, [&] { egr_build_task_order(c); }); // cached per (rank, world): flipping between two partitions costs nothing
}

int egr_set_grad_overwrite(egr_context *c, int enable) {
    if (!c) return 1;
    c->grad_overwrite = enable != 0;
    c->delta_pending = false;
    return 0;
}

int egr_grad_delta_consumed(egr_context *c) {
    if (!c) return 1;
    c->delta_pending = false;
    return 0;
}

int egr_set_exact_stats(egr_context *c, int enable) {
    if (!c) return 1;
    c->live_fresh = false;
    c->exact_stats = enable != 0; // the boxes change at the next egr_update_bvh / egr_rebuild_bvh; egr_raytrace checks that they did
    return 0;
}

int egr_set_rays_per_task(egr_context *c, int rays_per_task) {
    if (!c || !(rays_per_task == 0 || rays_per_task == 16 || rays_per_task == 32 || rays_per_task == 64)) return 1;
    c->rays_per_task = rays_per_task;
    return 0;
}

int egr_set_team_help(egr_context *c, int on) {
    if (!c || !(on == 0 || on == 1 || on == -1)) return 1;
    c->team_help = on;
    return 0;
}

int egr_set_strands(egr_context *c, int strands) {
    if (!c || strands < 1 || strands > c->strands) return 1;
    c->strands_active = strands;
    return 0;
}

static int require_ready(egr_context *c, bool need_bvh) {
    if (!c->bound || !c->have_gaussians) {
        c->last_error = "libegr_hip: egr_bind / egr_set_gaussians must be called first";
        return 1;
    }
    if (need_bvh && (!c->bvh_valid || c->n_built != c->g.count)) {
        c->last_error = "libegr_hip: BVH is missing or was built for a different gaussian count; call egr_rebuild_bvh";
        return 1;
    }
    return 0;
}

int egr_rebuild_bvh(egr_context *c, void *stream) {
    if (!c || require_ready(c, false)) return 1;
    return guarded(c, [&] { egr_bvh_rebuild(c, (hipStream_t)stream); });
}

int egr_update_bvh_ex(egr_context *c, unsigned flags, void *stream) {
    if (!c) return 1;
    c->live_fresh = false; // set again by a refit that wrote the live records (egr_bvh_refit), and only if it got that far
    if (require_ready(c, true)) return 1;
    if (flags & ~(unsigned)EGR_UPDATE_FUSE_LIVE) {
        c->last_error = "libegr_hip: egr_update_bvh_ex: unknown flag";
        return 1;
    }
    return guarded(c, [&] {
        hipStream_t s = (hipStream_t)stream;
        if (c->timing) EGR_HIP(hipEventRecord(c->ev_ub0, s));
        egr_bvh_refit(c, s, (flags & EGR_UPDATE_FUSE_LIVE) != 0);
        if (c->timing) EGR_HIP(hipEventRecord(c->ev_ub1, s)), c->have_ub = true;
    });
}
int egr_update_bvh(egr_context *c, void *stream) { return egr_update_bvh_ex(c, 0u, stream); }

int egr_raytrace(egr_context *c, int grads_enabled, void *stream) {
    if (!c) return 1;
    // The live records of an egr_update_bvh_ex(EGR_UPDATE_FUSE_LIVE) are honoured by the raytrace that is the NEXT call on this context
    // and by nothing else: the flag is consumed here, before anything can fail, so a launch that is refused (stale tree, exact-stats
    // gate) or any other call in between (egr_bind, egr_set_gaussians, egr_set_exact_stats, rebuild, another update) leaves the next
    // launch reading the live tensors itself.
    const bool live_fresh = c->live_fresh;
    c->live_fresh = false;
    if (require_ready(c, true)) return 1;
    if (c->exact_stats != c->boxes_are_cubes) {
        c->last_error = "libegr_hip: egr_set_exact_stats changed since the tree was last refitted; call egr_update_bvh or egr_rebuild_bvh first";
        return 1;
    }
    return guarded(c, [&] {
        hipStream_t s = (hipStream_t)stream;
        c->stamps_used = 0;
        if (c->timing) EGR_HIP(hipEventRecord(c->ev_rt0, s));
        egr_trace_launch(c, grads_enabled != 0, live_fresh, s);
        if (c->timing) EGR_HIP(hipEventRecord(c->ev_rt1, s)), c->have_rt = true;
    });
}

int egr_set_camera_from_dataset(egr_context *c, const float *rotation_c2w_dataset, const float *camera_center, float vertical_fov_radians, float znear, float zfar, void *stream) {
    if (!c || !c->bound || !rotation_c2w_dataset || !camera_center) return 1;
    return guarded(c, [&] { egr_set_camera_launch(c, rotation_c2w_dataset, camera_center, vertical_fov_radians, znear, zfar, (hipStream_t)stream); });
}

int egr_set_targets_chw(egr_context *c, const float *diffuse, const float *specular, const float *depth, const float *normal, const float *roughness, const float *f0, void *stream) {
    if (!c || !c->bound) return 1;
    const float *chw[6] = {diffuse, specular, depth, normal, roughness, f0};
    return guarded(c, [&] { egr_upload_targets(c, chw, (hipStream_t)stream); });
}

int egr_denoise(egr_context *c, void *stream) {
    if (!c || require_ready(c, false)) return 1;
    return guarded(c, [&] {
        if (c->denoise_mode == 0) egr_copy_final_to_denoised(c, (hipStream_t)stream); // EGR_DENOISE=0: plain copy
        else egr_denoise_atrous(c, (hipStream_t)stream);
    });
}

int egr_get_counters(egr_context *c, egr_counters *out, void *stream) { return egr_get_counters_ex(c, out, sizeof(egr_counters), stream); }

int egr_get_counters_ex(egr_context *c, void *out_raw, size_t out_bytes, void *stream) {
    if (!c || !out_raw) return 1;
    egr_counters full{}, *out = &full; // filled completely, then the caller gets as much of it as its struct holds
    const int rc = guarded(c, [&] {
        // the control block only crosses PCIe when somebody asks for it (nothing does inside a training iteration)
        EGR_HIP(hipMemcpyAsync(c->control_host, c->control, CW_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        EGR_HIP(hipStreamSynchronize((hipStream_t)stream));
        const uint32_t *w = c->control_host;
        auto u64 = [&](int i) { return (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32); };
        for (int s = 0; s < EGR_NSTEPS; s++)
            out->rays[s] = u64(CW_RAYS + 2 * s), out->candidates[s] = u64(CW_CAND + 2 * s), out->composited[s] = u64(CW_COMP + 2 * s), out->accepted[s] = u64(CW_ACCEPTED + 2 * s);
        out->lifetime_rays = u64(CW_LIFE_RAYS);
        out->lifetime_launches = w[CW_LIFE_LAUNCHES];
        out->status = w[CW_STATUS];
        out->bvh_depth = c->max_depth;
        out->bucket_records = w[CW_BUCKET_RECORDS];
        out->device_bytes = (uint64_t)c->device_bytes;
        out->arena_blocks_used = w[CW_HIT_BUMP], out->arena_blocks_cap = c->hit_blocks_cap;
        out->ext_blocks_used = w[CW_EXT_BUMP], out->ext_blocks_cap = c->ext_blocks_cap;
        if (getenv("EGR_DEBUG_EXT")) fprintf(stderr, "[egr] extension blocks used %u of %u, cand_cap %u\n", w[CW_EXT_BUMP], c->ext_blocks_cap, c->cand_cap);
        if (getenv("EGR_PRINT_TRAVERSAL_STATS")) {
            for (int k = 0; k < 2; k++)
                fprintf(stderr, "[egr stats %s] lane node visits %llu, lane leaf-box hits %llu, wave inner iterations %llu, wave outer rounds %llu\n", k ? "bounce" : "primary",
                        (unsigned long long)u64(CW_DBG + 8 * k), (unsigned long long)u64(CW_DBG + 8 * k + 2), (unsigned long long)u64(CW_DBG + 8 * k + 4), (unsigned long long)u64(CW_DBG + 8 * k + 6));
            for (int k = 0; k < 3; k++)
                fprintf(stderr, "[egr stats forward step %d] first wave exit -> last wave exit: %.3f ms\n", k,
                        (double)(u64(CW_DBG3 + 4 * k + 2) - u64(CW_DBG3 + 4 * k)) / 100e6 * 1e3);
            fprintf(stderr, "[egr stats team] offers made %u, offers walked by helpers %u (their walk batches: %u), owner walk batches that left >= EGR_DONATE_MIN pairs on the stack %u\n", c->control_host[CW_DBG3 + 12], c->control_host[CW_DBG3 + 13], c->control_host[CW_DBG3 + 14], c->control_host[CW_DBG3 + 15]);
            for (int k = 0; k < 2; k++)
                fprintf(stderr, "[egr stats backward %s] wave-cycles(s_memtime) per-hit math %llu, neighbour combine + LDS table %llu, wide adds %llu, table flush %llu; hit rows %llu\n",
                        k ? "bounce" : "primary", (unsigned long long)u64(CW_DBG2 + 16 + 10 * k), (unsigned long long)u64(CW_DBG2 + 18 + 10 * k),
                        (unsigned long long)u64(CW_DBG2 + 20 + 10 * k), (unsigned long long)u64(CW_DBG2 + 22 + 10 * k), (unsigned long long)u64(CW_DBG2 + 24 + 10 * k));
            fprintf(stderr, "[egr stats forward chain] wave-cycles(s_memtime) whole chains (task pull to end) %llu, of which step epilogues %llu\n", (unsigned long long)u64(CW_DBG2 + 12), (unsigned long long)u64(CW_DBG2 + 8));
            fprintf(stderr, "[egr stats primary composite] wave-cycles(s_memtime) selection scans %llu, arena block %llu, (alpha, record) fetch %llu, pass 1 %llu, appearance pass %llu\n", (unsigned long long)u64(CW_DBG2 + 40), (unsigned long long)u64(CW_DBG2 + 42), (unsigned long long)u64(CW_DBG2 + 44), (unsigned long long)u64(CW_DBG2 + 46), (unsigned long long)u64(CW_DBG2 + 48));
            fprintf(stderr, "[egr stats primary lists] tiles by their longest candidate list: <=16: %u, <=24: %u, <=32: %u, <=40: %u, <=48: %u, <=64: %u, longer: %u\n", w[CW_DBG2 + 50], w[CW_DBG2 + 51], w[CW_DBG2 + 52], w[CW_DBG2 + 53], w[CW_DBG2 + 54], w[CW_DBG2 + 55], w[CW_DBG2 + 56]);
            fprintf(stderr, "[egr stats primary leaf filter] leaves before the sphere / pyramid test %llu, wave-cycles in the test %llu\n", (unsigned long long)u64(CW_DBG2 + 38), (unsigned long long)u64(CW_DBG2 + 36));
            for (int k = 0; k < 2; k++)
                fprintf(stderr, "[egr stats %s] wave-cycles(s_memtime) traversal %llu composite %llu | of the traversal: leaf evaluation (frustum walk) %llu (slot +8: %llu)\n", k ? "bounce" : "primary",
                        (unsigned long long)u64(CW_DBG2 + 4 * k), (unsigned long long)u64(CW_DBG2 + 4 * k + 2), (unsigned long long)u64(CW_DBG2 + 8 + 4 * k + 2),
                        (unsigned long long)u64(CW_DBG2 + 8 + 4 * k));
        }
    });
    if (rc == 0) memcpy(out_raw, &full, out_bytes < sizeof(full) ? out_bytes : sizeof(full));
    return rc;
}

int egr_reset_lifetime_counters(egr_context *c, void *stream) {
    if (!c) return 1;
    return guarded(c, [&] { EGR_HIP(hipMemsetAsync(c->control + CW_LIFE_RAYS, 0, 4 * sizeof(uint32_t), (hipStream_t)stream)); });
}

int egr_enable_timing(egr_context *c, int enable) {
    if (!c) return 1;
    c->timing = enable != 0;
    c->have_rt = c->have_ub = false;
    c->stamps_used = 0;
    return 0;
}
float egr_last_raytrace_ms(egr_context *c) {
    if (!c || !c->have_rt) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev_rt1) != hipSuccess || hipEventElapsedTime(&ms, c->ev_rt0, c->ev_rt1) != hipSuccess) return -1.0f;
    return ms;
}
float egr_last_update_bvh_ms(egr_context *c) {
    if (!c || !c->have_ub) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev_ub1) != hipSuccess || hipEventElapsedTime(&ms, c->ev_ub0, c->ev_ub1) != hipSuccess) return -1.0f;
    return ms;
}
int egr_last_kernel_ms(egr_context *c, float *ms, const char **names, int max_entries) {
    if (!c) return 0;
    int n = 0;
    for (size_t i = 0; i < c->stamps_used && n < max_entries; i++) {
        float t = -1.0f;
        if (hipEventSynchronize(c->stamps[i].stop) != hipSuccess) break;
        if (hipEventElapsedTime(&t, c->stamps[i].start, c->stamps[i].stop) != hipSuccess) break;
        ms[n] = t;
        if (names) names[n] = c->stamps[i].name;
        n++;
    }
    return n;
}

int egr_debug_get_instances(egr_context *c, float *M, float *W, float *aabb, void *stream) {
    if (!c) return 1;
    return guarded(c, [&] {
        EGR_HIP(hipStreamSynchronize((hipStream_t)stream));
        size_t n = c->n_built;
        std::vector<uint32_t> pos(n);
        std::vector<float> tmp(n * 16);
        EGR_HIP(hipMemcpy(pos.data(), c->pos_of_gid, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        auto unpermute = [&](const float4 *src, float *dst, size_t stride) { // records live at sorted positions; report them per gaussian id
            EGR_HIP(hipMemcpy(tmp.data(), src, n * stride * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; i++) memcpy(dst + 12 * i, tmp.data() + stride * (size_t)pos[i], 12 * sizeof(float));
        };
        if (M) { // backward records hold (M row, exp(scale)); the reported 3x4 transform carries the mean in column 3
            unpermute(c->inst_m, M, 16);
            std::vector<float> mean(3 * n);
            EGR_HIP(hipMemcpy(mean.data(), c->g.mean, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; i++)
                for (int a = 0; a < 3; a++) M[12 * i + 4 * a + 3] = mean[3 * i + a];
        }
        if (W) unpermute(c->inst_w, W, 16);
        if (aabb) EGR_HIP(hipMemcpy(aabb, c->aabb, n * 6 * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int egr_debug_set_pixel_mask(egr_context *c, const uint8_t *device_mask) {
    if (!c) return 1;
    c->pixel_mask = device_mask;
    return 0;
}

int egr_debug_get_step_hits(egr_context *c, int32_t *host_out, void *stream) {
    if (!c || !host_out || require_ready(c, false)) return 1;
    return guarded(c, [&] { egr_export_step_hits(c, host_out, (hipStream_t)stream); });
}

int egr_debug_get_hit_sequence_hash(egr_context *c, uint64_t *host_out, void *stream) {
    if (!c || !host_out || require_ready(c, false)) return 1;
    return guarded(c, [&] { egr_export_hit_hash(c, host_out, (hipStream_t)stream); });
}

// Unit test hook for the division / square root of the hot arithmetic (egr_device.hpp): quot[i] = egr_div_rn(a[i], b[i], egr_rcp_refined(b[i])), root[i] = egr_sqrt_rn(a[i]).
__global__ void k_debug_lean_arith(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ quot, float *__restrict__ root, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    quot[i] = egr_div_rn(a[i], b[i], egr_rcp_refined(b[i]));
    root[i] = egr_sqrt_rn(a[i]);
}
int egr_debug_lean_arith(int device, const float *a, const float *b, float *quot, float *root, uint32_t n, void *stream) {
    if (!a || !b || !quot || !root) return 1;
    if (hipSetDevice(device) != hipSuccess) return 1;
    if (n) hipLaunchKernelGGL(k_debug_lean_arith, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a, b, quot, root, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int egr_debug_check_bvh(egr_context *c, void *stream) {
    if (!c) return 1;
    int bad = 0;
    int rc = guarded(c, [&] {
        std::string msg;
        bad = egr_bvh_check(c, (hipStream_t)stream, msg);
        if (bad) c->last_error = "libegr_hip: BVH check failed: " + msg;
    });
    return rc ? rc : bad;
}

const char *egr_last_error(egr_context *c) { return c ? c->last_error.c_str() : "libegr_hip: null context"; }

} // extern "C"
