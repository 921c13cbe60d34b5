/*
 * egr_raytracer.h -- C ABI of the MI355X-native differentiable Gaussian ray tracer (libegr_hip.so).
 *
 * Drop-in boundary for the hot path of graphdeco-inria/editable-gaussian-reflections:
 * editable_gauss_refl/cuda (OptiX/CUDA). The reference exposes this path as TorchScript custom classes
 * (TORCH_LIBRARY(raytracer, m), cuda/csrc/raytracer.cpp:208-218); every entry point below replaces one
 * method of that `Raytracer` class and takes exactly the raw device pointers its `reify()` structs hold.
 * No torch types cross this boundary. The thin TORCH_LIBRARY shim that re-exports these functions under
 * the reference's class names lives in editable-gaussian-reflections_amd/csrc/torch_binding.cpp; the
 * reference-side binding is shown in INTEGRATION.md.
 *
 * All pointers are DEVICE pointers (HIP, gfx950) unless stated otherwise. All arithmetic is fp32.
 * Functions return 0 on success, non-zero on failure; egr_last_error() gives the message. Nothing here
 * synchronises the stream unless documented. Not thread-safe per context (the reference is not either:
 * train.py:214-215 serialises callers with a lock).
 *
 * Paths in comments are relative to /root/reference/editable_gauss_refl/cuda/csrc/.
 */
#ifndef EGR_RAYTRACER_H
#define EGR_RAYTRACER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGR_MAX_BOUNCES 2                       /* flags.h:4  */
#define EGR_NUM_STEPS (EGR_MAX_BOUNCES + 1)
#define EGR_MAX_ALPHA 0.9999f                   /* flags.h:7  */
#define EGR_ROUGHNESS_DOWNWEIGHT_GRAD 1         /* flags.h:11 */
#define EGR_ROUGHNESS_DOWNWEIGHT_GRAD_POWER 3.0f /* flags.h:12 */
#define EGR_MAX_COMPOSITED_PER_RAY (16 * 99)    /* BUFFER_SIZE * MAX_ITERATIONS, flags.h:15-16 */
#define EGR_PPLL_NULL_PTR (2u << 29)            /* core/per_pixel_linked_list.h:4 */

typedef struct egr_context egr_context; /* replaces struct Raytracer (raytracer.cpp:24-79) */

/* core/gaussians.h:3-25 -- raw (pre-activation) parameters, row-major [count, k], and their gradients */
typedef struct egr_gaussians {
    uint32_t count;
    const float *rgb;       /* [N,3] relu            */
    const float *normal;    /* [N,3] identity        */
    const float *f0;        /* [N,3] clip01          */
    const float *roughness; /* [N,1] clip01          */
    const float *opacity;   /* [N,1] sigmoid         */
    const float *scale;     /* [N,3] exp             */
    const float *mean;      /* [N,3] identity        */
    const float *rotation;  /* [N,4] normalise (r,x,y,z) */
    float *dL_drgb, *dL_dnormal, *dL_df0, *dL_droughness, *dL_dopacity, *dL_dscale, *dL_dmean, *dL_drotation;
    float *total_weight;    /* [N,1] */
} egr_gaussians;

/* core/config.h:5-26 -- device-resident scalars, read by pointer inside the kernels (Python mutates them in place) */
typedef struct egr_config {
    const float *exp_power, *alpha_threshold, *transmittance_threshold;
    const uint8_t *accumulate_samples, *jitter_primary_rays; /* torch bool */
    const int32_t *num_bounces;
    const float *global_scale_factor;
    const float *loss_weight_diffuse, *loss_weight_specular, *loss_weight_depth, *loss_weight_normal, *loss_weight_f0,
        *loss_weight_roughness;
    const float *eps_forward_normalization, *eps_scale_grad, *eps_ray_surface_offset, *eps_min_roughness;
    const float *reflection_invalid_normal_threshold, *backfacing_invalid_normal_threshold, *backfacing_max_dist;
} egr_config;

/* core/camera.h:8-15 */
typedef struct egr_camera {
    const float *origin;               /* [3]   */
    const float *vertical_fov_radians; /* [1]   */
    const float *rotation_c2w;         /* [3,3] */
    const float *rotation_w2c;         /* [3,3] = c2w^T (camera.h:67) */
    const float *znear, *zfar;         /* [1]   */
} egr_camera;

/* core/framebuffer.h:72-102 -- outputs [3,H,W,c] indexed pixel_id + H*W*step (:132), final/denoised [1,H,W,3] */
typedef struct egr_framebuffer {
    float *output_rgb, *output_depth, *output_normal, *output_f0, *output_roughness, *output_transmittance,
        *output_total_transmittance, *output_ray_origin, *output_ray_direction, *output_final, *output_denoised;
    float *accumulated_rgb, *accumulated_transmittance, *accumulated_total_transmittance, *accumulated_depth,
        *accumulated_normal, *accumulated_f0, *accumulated_roughness;
    int32_t *accumulated_sample_count; /* [1] */
    const float *target_diffuse, *target_specular, *target_depth, *target_normal, *target_f0, *target_roughness;
} egr_framebuffer;

/* core/metadata.h:3-7 */
typedef struct egr_metadata {
    uint8_t *grads_enabled;    /* [1] torch bool, written by egr_raytrace from its argument (metadata.h:29) */
    int32_t *total_num_calls;  /* [1] incremented by egr_raytrace before the launch (metadata.h:30)        */
    int32_t *random_seeds;     /* [H,W,1] final per-pixel RNG state (shaders.cu:172)                        */
} egr_metadata;

/* core/stats.h:3-6 */
typedef struct egr_stats {
    int32_t *num_accumulated_per_pixel; /* [H,W] composited hits of the LAST executed step (forward_pass.cu:140) */
    int32_t *num_traversed_per_pixel;   /* [H,W] intersection-program invocations, all steps (forward_pass.cu:46) with egr_set_exact_stats;
                                         * by default the subset of them whose response point lies inside the gaussian's ellipsoid (see there) */
} egr_stats;

/* Whole-launch work counters (not in the reference; used for the roofline's algorithmic bytes). */
typedef struct egr_counters {
    uint64_t rays[EGR_NUM_STEPS];        /* rays traced per bounce step (a ray = one pixel x step)            */
    uint64_t candidates[EGR_NUM_STEPS];  /* candidates per step: those inside their ellipsoid (default), or Hc = cube overlaps with exact stats */
    uint64_t composited[EGR_NUM_STEPS];  /* composited hits per step (Kc)                                     */
    uint64_t lifetime_rays;              /* rays of ALL launches since egr_create / egr_reset_lifetime_counters */
    uint32_t lifetime_launches;
    uint32_t status;                     /* EGR_STATUS_* bit mask of the last launch                          */
    uint32_t bvh_depth;
    uint32_t bucket_records;             /* 64-B gradient records (16-lane atomic adds) the backward chain sent to the gradient rows in this launch */
    uint64_t device_bytes;               /* device memory this context holds right now (scratch, arena, ray state, tree, records);
                                          * the caller's tensors (parameters, gradients, framebuffer) are not included          */
    uint32_t arena_blocks_used, arena_blocks_cap; /* composited-hit arena (backward capacity): 9-KB blocks the last grad launch took (waves take them in runs of 8:
                                                   * up to 7 per resident wave are taken and not written) / holds */
    uint32_t ext_blocks_used, ext_blocks_cap;     /* candidate-list extension blocks (forward capacity) the last launch took / holds         */
    uint64_t accepted[EGR_NUM_STEPS];    /* accepted candidates per step = entries the reference inserts into its forward list
                                          * (shaders.cu:74; its counter runs on over the three steps, so their SUM must stay below
                                          * ppll_forward_size upstream - there is no check there)                               */
} egr_counters;

#define EGR_STATUS_OK 0u
#define EGR_STATUS_CANDIDATE_OVERFLOW 1u /* a ray met more candidates than the forward capacity allows (dropped) */
#define EGR_STATUS_HIT_ARENA_OVERFLOW 2u /* composited-hit arena (backward capacity) exhausted; gradients partial */

/* Raytracer::Raytracer(width, height, num_gaussians, ppll_forward_size, ppll_backward_size) (raytracer.cpp:45-79).
 * The two sizes are entry counts of the reference's per-pixel linked lists (36 B per entry,
 * per_pixel_linked_list.h:6-16); this implementation spends the same byte budgets on its candidate scratch and
 * its composited-hit arena. device = HIP device ordinal the buffers live on. */
int egr_create(egr_context **ctx, int device, int width, int height, int64_t ppll_forward_size, int64_t ppll_backward_size);
void egr_destroy(egr_context *ctx);

/* params_on_host.{camera,config,framebuffer,metadata,stats} = holder->reify() (raytracer.cpp:61-68) */
int egr_bind(egr_context *ctx, const egr_camera *camera, const egr_config *config, const egr_framebuffer *framebuffer,
             const egr_metadata *metadata, const egr_stats *stats);

/* Raytracer::resize re-reify + upload of Params.gaussians (raytracer.cpp:112-120) */
int egr_set_gaussians(egr_context *ctx, const egr_gaussians *gaussians);

/* Raytracer::rebuild_bvh (raytracer.cpp:102-110; optix/bvh_wrapper.h:24-30,118-157): instance transforms from
 * the current parameters + full LBVH build. Synchronises the stream (the reference's build also does). */
int egr_rebuild_bvh(egr_context *ctx, void *hip_stream); /* fails for count >= 2^26 (67M): record indices are packed into 26 bits */

/* Raytracer::update_bvh (raytracer.cpp:100; optix/bvh_wrapper.h:32-59): re-snapshot instance transforms and
 * refit the existing tree. Asynchronous; reads alpha_threshold/exp_power/global_scale_factor on the device
 * (the reference does three blocking .item() reads, bvh_wrapper.h:38-40). */
int egr_update_bvh(egr_context *ctx, void *hip_stream);
/* The same with flags. EGR_UPDATE_FUSE_LIVE: the pass over the cloud that snapshots the transforms also writes the LIVE per-gaussian
 * records (activated appearance, opacity, sigma: what the reference's read_* helpers fetch inside the launch, utils/helpers.cu:10-33),
 * and the NEXT egr_raytrace does not repeat that pass. The caller promises that the parameter tensors and the config scalars the live
 * records depend on (alpha_threshold, exp_power) are not written between this call and that egr_raytrace (the reference's caller, gaussian_raytracer.py:139-142, calls the two back to back); every later launch
 * reads the live parameters again as usual. One pass of ~130 B per gaussian less per training iteration. */
#define EGR_UPDATE_FUSE_LIVE 1u
int egr_update_bvh_ex(egr_context *ctx, unsigned flags, void *hip_stream);

/* Raytracer::raytrace (raytracer.cpp:81-94): metadata update, stats reset, one launch of the whole
 * forward (+ backward when grads_enabled) path = __raygen__rg (shaders.cu:77-173), then
 * accumulated_sample_count += 1 when accumulate_samples. grads_enabled is what
 * torch::autograd::GradMode::is_enabled() returned in the caller (metadata.h:29). Asynchronous. */
int egr_raytrace(egr_context *ctx, int grads_enabled, void *hip_stream);

/* Camera upload (replaces the caller's ten tiny tensor kernels of renderer/gaussian_raytracer.py:94-100, which stay valid): rotation_c2w_dataset =
 * the dataset's camera-to-world rotation `viewpoint_camera.R` (row-major 3x3 fp32), camera_center = `viewpoint_camera.camera_center` (3 fp32), both DEVICE
 * pointers. ONE launch writes the bound camera struct: rotation_c2w = -R with column 0 negated back (the reference's Blender convention), its transpose,
 * the origin, and the three scalars. Asynchronous on the stream. */
int egr_set_camera_from_dataset(egr_context *ctx, const float *rotation_c2w_dataset, const float *camera_center, float vertical_fov_radians, float znear,
                                float zfar, void *hip_stream);

/* Target upload (replaces the caller's six `framebuffer.target_*.copy_(image.moveaxis(0, -1))` of renderer/gaussian_raytracer.py:109-137, which
 * stay valid): device pointers to channel-major fp32 images ([3][H][W] diffuse, specular, normal, f0; [1][H][W] depth, roughness), NULL = the
 * target is absent and reads zero (the reference zeroes the buffer). ONE launch writes the framebuffer's pixel-major target buffers for the pixels
 * of THIS context's partition only - the only target pixels its launches read; the target_* buffers OUTSIDE the partition's tiles keep whatever they held
 * (undefined for a caller that reads them; a context that is switched to another partition or to the whole image uploads its targets again). Asynchronous on the stream. */
int egr_set_targets_chw(egr_context *ctx, const float *diffuse, const float *specular, const float *depth, const float *normal, const float *roughness,
                        const float *f0, void *hip_stream);

/* Raytracer::denoise (raytracer.cpp:96; optix/denoiser_wrapper.h:42-105: HDR image output_final + normal guide output_normal
 * -> output_denoised). The OptiX AI denoiser is a closed network; the stand-in is an edge-avoiding a-trous wavelet filter with
 * the same inputs and output (csrc/denoise.hip; parity with OptiX is unpinned). Env EGR_DENOISE=0 at creation: plain copy. */
int egr_denoise(egr_context *ctx, void *hip_stream);

/* Multi-GPU image partition (not in the reference): this context only traces the 16x16-pixel macro tiles it owns;
 * default rank 0 of 1 = whole image. Ownership: the macro tiles (index m = my * ceil(width / 16) + mx) are sorted along a Z-curve over
 * (mx, my); the tile at position i of that order belongs to rank (i + i / world_size) % world_size - every run of world_size
 * consecutive positions (a compact 2-D block of the image) holds each rank once, rotated by the block index. Hosts that need the map
 * (image gathers, pixel masks) ask egr_tile_owner instead of re-implementing it. */
int egr_set_partition(egr_context *ctx, int rank, int world_size);
/* The rank that owns macro tile `tile_index` (= my * ceil(width / 16) + mx) of a width x height image cut `world_size` ways; -1 for
 * arguments out of range. Pure host function (no context, no device). */
int egr_tile_owner(int width, int height, int world_size, int tile_index);

/* Per-launch gradient buffers (not in the reference; the multi-GPU exchange step). The reference's launch ADDS to the
 * gradient tensors (atomicAdd, backward_pass.cu:210-220) and so does this library by default. With enable != 0 the dL_d* / total_weight
 * pointers of egr_set_gaussians are taken as a PER-LAUNCH buffer: the first grad launch after the caller has consumed the buffer
 * (egr_grad_delta_consumed; also right after this call) STORES its sums there (rows no ray touched read 0), so the caller can all-reduce
 * the buffer over the ranks and add it to its persistent gradients without clearing it in between (one write pass over [22N] instead of
 * a read-modify-write, and no memset per iteration). A grad launch that finds the buffer NOT consumed yet - two launches before one
 * fold: multi-view accumulation, a caller that raised between launch and fold - ADDS to it like the default path, so no launch's
 * gradients or total_weight are ever dropped. */
int egr_set_grad_overwrite(egr_context *ctx, int enable);
/* The caller has folded the per-launch buffer into its persistent gradients (after the all-reduce): the next grad launch stores again.
 * REQUIRED after every fold (changed in library version 0.6: until then every launch stored): a host that all-reduces the buffer in place and
 * never calls this has every later launch ADD to the already rank-summed values, and the next reduce sums those again. */
int egr_grad_delta_consumed(egr_context *ctx);

/* Exact statistics (not in the reference). By default the tree bounds each Gaussian's ELLIPSOID and the walk only has to find the
 * instances whose response point can be accepted, so stats.num_traversed_per_pixel and egr_counters.candidates count the candidates
 * whose response point lies INSIDE the ellipsoid (|u|^2 <= 1, shaders.cu:48) on the ray's segment - accepted ones plus the back-face /
 * behind-the-origin rejections among them: a property of ray and gaussian, whatever the walk, and pixel by pixel a SUBSET of the reference's
 * intersection-program invocations (every instance whose cube the segment overlaps, shaders.cu:33; about half of them on surfel
 * clouds); every other output is unaffected.
 * With enable != 0 the next egr_update_bvh / egr_rebuild_bvh bounds the instance CUBES (what OptiX's TLAS holds) and every launch
 * counts exactly the instances whose unit cube the ray segment overlaps: the reference's number, at a slower walk. egr_raytrace
 * fails if the flag changed since the last refit. Used by tests and by bench.py to measure Hc. */
int egr_set_exact_stats(egr_context *ctx, int enable);

/* Strands (not in the reference): egr_raytrace cuts this context's tiles into `strands` slices and runs their step kernels on
 * separate internal HIP streams (forked from / joined to the caller's stream), so that one slice's persistent-wave tail is
 * filled by another slice's next kernel. 1 = everything on the caller's stream (per-kernel timings are then exclusive).
 * Accepts 1..(value at creation: env EGR_STRANDS, default 3); returns 1 otherwise. Without a call a launch uses ONE strand (round 5:
 * with today's chains three strands measure 1-3 % slower than one; the mechanism stays for experiments). */
int egr_set_strands(egr_context *ctx, int strands);

/* Task shape (not in the reference): pixels one wave traces together. 64 = 8x8 (default), 32 = 8x4, 16 = 4x4; 0 = automatic (64, or
 * 32 for a rank of a partition with fewer than two 8x8 tiles per resident wave when team help is off). Also settable at creation with env
 * EGR_RAYS_PER_TASK. The ORDER of exactly tied hits of a bounce ray depends on the shape (deviation (a)); parity tests
 * that trace single macro tiles through egr_set_partition pin the shape of the run they compare with. Returns 1 for other values. */
int egr_set_rays_per_task(egr_context *ctx, int rays_per_task);

/* Team help (not in the reference): the forward chain's workgroups are teams of waves with a shared LDS; with help on, a wave that has no tile
 * left (or waits for its own helpers) walks (ray, node) pairs that a team mate with a long pair stack puts on offer - several waves on one heavy
 * tile: the tail of every launch, and most of the launch of a rank of a multi-GPU partition. Help changes the ORDER in which a ray's candidates
 * enter its list, never the set; that order reaches an output only where two candidates of a ray have EXACTLY the same distance (the depth
 * selection orders by (t, list index); the total transmittance is an fp64 product rounded once): without exact ties the images of a run with
 * help equal those of a run without bit for bit (tests/test_hip_parity.py), with them the tied hits may composite in another order from run to
 * run (deviation (a); upstream the order of tied hits is its PPLL's insertion order, which is timing too). 1 = on (default since
 * round 5: whole image -6 % / -3 % forward chain), 0 = off (launches reproducible bit for bit), -1 = on only for a rank of a partition with fewer than
 * two 8x8 tiles per wave slot; also env EGR_TEAM_HELP (0 / 1) at creation. Returns 1 for other values. (1 also selects the backward chain's team build, which an under-filled rank of a partition gets in any
 * case: its waves without tiles take batches of their team mates' bounce hits - gradients are atomic adds, no result depends on it.) */
int egr_set_team_help(egr_context *ctx, int on);

/* Synchronises the stream and returns the work counters / status of the most recent egr_raytrace.
 * ABI: egr_counters only ever GROWS AT ITS END (version string of egr_version() bumps with it). egr_get_counters writes
 * sizeof(egr_counters) of THIS header; a host compiled against an older header passes its own sizeof to egr_get_counters_ex, which
 * writes min(out_bytes, sizeof(egr_counters)) bytes - never more than the caller's struct holds. */
int egr_get_counters(egr_context *ctx, egr_counters *out, void *hip_stream);
int egr_get_counters_ex(egr_context *ctx, void *out, size_t out_bytes, void *hip_stream);
int egr_reset_lifetime_counters(egr_context *ctx, void *hip_stream);

/* Wall-clock of the last egr_raytrace / egr_update_bvh on the GPU (HIP events recorded on the launch stream
 * when timing is enabled). Returns milliseconds, <0 if not available. Synchronises on the end event. */
int egr_enable_timing(egr_context *ctx, int enable);
float egr_last_raytrace_ms(egr_context *ctx);
float egr_last_update_bvh_ms(egr_context *ctx);
/* duration of the individual kernels of the last egr_raytrace, in launch order; returns count written */
int egr_last_kernel_ms(egr_context *ctx, float *ms, const char **names, int max_entries);

/* Copies the snapshot instance records for parity tests: M[N*12], W[N*12] (3x4 row-major), aabb[N*6]
 * (lo xyz, hi xyz; invisible instances have lo > hi). Host pointers; synchronises. */
int egr_debug_get_instances(egr_context *ctx, float *M, float *W, float *aabb, void *hip_stream);
/* Composited hits per pixel and bounce step of the last egr_raytrace with grads_enabled (what backward_pass.cu:38-45 calls
 * num_hits[step]; the reference keeps it in registers, stats.num_accumulated_per_pixel only shows the last step): host_out =
 * int32 [EGR_NUM_STEPS][H*W], host pointer; pixels outside this context's partition read 0. Parity tests use it to LIST the
 * pixels whose bounce rays met a different number of hits than the CPU oracle's. Synchronises. */
int egr_debug_get_step_hits(egr_context *ctx, int32_t *host_out, void *hip_stream);
/* The ORDERED sequence of gaussians every pixel composited on every step of the last egr_raytrace with grads_enabled, as a hash: host_out = uint64
 * [EGR_NUM_STEPS][H*W] (host pointer), sum_i (id_i + 1) B^i mod 2^64 over the composite index i (front to back), B = 0x9E3779B97F4A7C15; 0 = no hit, and for
 * pixels outside this context's partition. The CPU oracle reports the same number (oracle/egr_oracle.cpp: Outputs::hit_sequence_hash): a pixel composited the
 * same hits in the same order on both sides iff the hashes agree - the definition of a "clean" pixel in the at-size gradient check. Synchronises. */
int egr_debug_get_hit_sequence_hash(egr_context *ctx, uint64_t *host_out, void *hip_stream);
/* Pixel mask for parity tests: device_mask = uint8 [H*W] in DEVICE memory (caller-owned, must outlive the launches that use it), or NULL
 * to clear. A pixel whose mask byte is 0 is treated like a pixel outside the image by every kernel of a launch: no ray, no outputs written,
 * no statistics, no gradient contribution. The CPU oracle has the same hook (orc_set_pixel_mask), so both sides can trace exactly the pixels
 * whose per-step hit counts agree (tests/test_hip_configs.py: the at-size gradient check). Takes effect at the next egr_raytrace. */
int egr_debug_set_pixel_mask(egr_context *ctx, const uint8_t *device_mask);
/* Unit-test hook: the division and square root of the hot per-candidate / per-hit arithmetic (csrc/egr_device.hpp: the compiler's IEEE correction steps without
 * its range scaling). quot[i] = a[i] / b[i], root[i] = sqrt(a[i]) as THOSE functions compute them; device pointers, n elements each. ACCEPTED DOMAIN: the results
 * are the correctly rounded ones whenever 2^-100 <= |b| <= 2^100, |a / b| is a normal number or 0, and a is 0 or a normal number >= 2^-100 for the root
 * (tests/test_hip_parity.py holds them to `/` and sqrtf bit for bit over that domain); outside it - b = 0, b = inf, a = inf, denormal radicands - they return NaN
 * or a result that is off in the last bits where IEEE gives inf / 0 / a denormal: a degenerate gaussian (all-zero quaternion, |W d| out of range) then drops out of
 * a ray's list through the NaN comparisons instead of contributing an inf. Asynchronous on the stream. */
int egr_debug_lean_arith(int device, const float *a, const float *b, float *quot, float *root, uint32_t n, void *hip_stream);
/* BVH self-check: every leaf box equals its instance box, every internal box is the union of its children,
 * every visible instance is reachable exactly once. Returns 0 if consistent. Host-side; synchronises. */
int egr_debug_check_bvh(egr_context *ctx, void *hip_stream);

const char *egr_last_error(egr_context *ctx);
const char *egr_version(void);

/* ---- `simple_knn._C.distCUDA2` (editable_gauss_refl/scene/gaussian_model.py:17, called at :197-201 and
 * :246-250 to initialise the scales). out[i] = mean of the squared distances from point i to its 3 nearest OTHER points
 * (exact; duplicates count with distance 0). points_xyz = [n][3] fp32, out = [n] fp32, both device pointers. Synchronises the
 * stream before returning (it frees its temporaries). Returns 0 on success; egr_knn_last_error() describes a failure. */
int egr_knn_mean_dist2(int device, const float *points_xyz, uint32_t n, float *out_mean_dist2, void *hip_stream);
const char *egr_knn_last_error(void);

/* ---- the host step around every raytrace as ONE launch - gradient import
 * (renderer/gaussian_raytracer.py:50-58), scale decay (train.py:224-226), torch.optim.Adam(eps=1e-15) over the 8 parameter
 * groups (scene/gaussian_model.py:296-338), clamps (train.py:251-254), both zero_grads (train.py:248-249) and the parameter
 * export of the next iteration (gaussian_raytracer.py:36-48). All pointers are device pointers to [n][width] fp32 arrays. */
#define EGR_MAX_PARAM_GROUPS 8
typedef struct egr_param_group {
    float *param;       /* model parameter, updated in place                                              */
    float *grad;        /* model gradient: read, then zeroed (NULL: none)                                   */
    float *rt_param;    /* raytracer-side tensor refreshed with the new value (export; NULL: skip)          */
    float *rt_grad;     /* raytracer-side gradient dL_d*: added to the gradient, then zeroed (NULL: skip)   */
    float *exp_avg;     /* Adam first moment  (NULL together with exp_avg_sq: no optimizer update)          */
    float *exp_avg_sq;  /* Adam second moment                                                               */
    uint32_t width;     /* floats per gaussian                                                               */
    float lr;           /* this iteration's learning rate of the group                                      */
    float clamp_min, clamp_max; /* applied after the update (-INFINITY / +INFINITY: none)                    */
    float log_decay;    /* != 1: param = log(exp(param) * log_decay) before the update (scale decay)        */
    uint32_t step;      /* this group's own 1-based Adam step count; 0: use the `step` argument             */
} egr_param_group;
/* `step` is Adam's 1-based step count of THIS update (torch keeps one per parameter tensor: a group whose state was re-created
 * - gaussian_model.py replace_tensor_to_optimizer - passes its own count in egr_param_group.step). Asynchronous on the stream. Returns 0 on success. */
int egr_fused_adam_step(int device, const egr_param_group *groups, int num_groups, uint32_t n, uint32_t step, double beta1, double beta2,
                        double eps, void *hip_stream);
const char *egr_fused_step_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* EGR_RAYTRACER_H */
