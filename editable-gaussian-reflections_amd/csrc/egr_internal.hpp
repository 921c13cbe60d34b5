// Internal declarations shared by the HIP translation units of libegr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/egr_raytracer.h"

#define EGR_NSTEPS EGR_NUM_STEPS
#define EGR_WAVE 64
#define EGR_TILE 8            // one wave = one 8x8 pixel tile
#define EGR_MACRO_TILE 16     // multi-GPU partition granule (2x2 wave tiles)
#define EGR_INTERNAL_NODE 0xFFFFFFFFu
#define EGR_LEAF_FLAG 0x80000000u  // child slot link: leaf -> EGR_LEAF_FLAG | record index, internal -> child node index
#define EGR_EMPTY_SLOT 0xFFFFFFFFu // unused child slot (checked before the leaf flag)
#define EGR_WIDTH 8                // children per wide node: 8 x 16 B = one 128-B cache line
#define EGR_MAX_STRANDS 4
#ifndef EGR_QUEUE_STRIDE
#define EGR_QUEUE_STRIDE 32u         // words between two task queue heads: every head on a 128-B line of its own (eight heads on ONE line = eight queues behind one atomic unit)
#endif
#define EGR_QUEUE_WORDS (16u * EGR_QUEUE_STRIDE) // task queue heads per strand: 2 kernels (forward chain, backward chain) x 8 XCD heads
#define EGR_GSTK 232               // x 64 = entries of a resident wave's global spill column for its (ray, node) pair stack (the first EGR_PSTK live in LDS)
#define EGR_EXT_BLOCK 16384u // entries of one candidate-list extension block
#define EGR_EXT_NONE 0xFFFFFFFFu
#define EGR_HIT_BLOCK_ROWS 8  // composited-hit arena block: 8 rows x 64 lanes x 16 B (+1 header row)
#define EGR_MAX_DEPTH_BINS 256

struct EgrCheck {
    hipError_t e;
    const char *what;
};
inline void egr_hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw EgrCheck{e, what};
}
#define EGR_HIP(expr) egr_hip_check((expr), #expr)

// ---- per-Gaussian records (internal layout in HBM) -----------------------------------------------------
// inst_w : float4[4N]   64-B test record: rows of W = M^-1 (world->object; snapshot at update/rebuild) + live
//                       quarter (f0.z, roughness, opacity, sigma) written per launch: one candidate test = one 64-B sector
// inst_m : float4[4N]   64-B backward record: rows 0-2 = (row of M (object->world; snapshot), exp(scale_a)), row 3 = raw quaternion;
//                       exp(scale) and the quaternion are LIVE: k_live rewrites them in every grad launch (backward_pass.cu:68-70
//                       reads them from the parameter tensors), so between an update and the next grad launch the record mixes
//                       snapshot M rows with the scale / rotation of the last grad launch
// grad_rows: float[32N] gradient accumulation, one 128-B line per gaussian (22 components used), zero between launches
// bsph   : float4[N]    bounding sphere of the ellipsoid (centre xyz, squared radius): 16 B, snapshot like W
// app    : float4[2N]   live per-launch record: (relu rgb, n.x) (n.y, n.z, f0.x, f0.y)       32 B
// wnodes : uint4[8*Nw]     8-wide BVH, one 128-B line per node; child slot (16 B):
//          x = lo.x | lo.y<<16, y = lo.z | hi.x<<16, z = hi.y | hi.z<<16 (16-bit cells of the build frame), w = link
// inst_w / inst_m / app / grad_rows are indexed by SORTED POSITION (Morton / leaf order);
// gid_of_pos / pos_of_gid map between sorted positions and the caller's gaussian ids

struct BvhFrame { // quantisation frame of one build: cell = (x - o) * s + 2
    float ox, oy, oz, sx, sy, sz;
};

struct DeviceView { // everything a kernel needs, passed by value
    int width, height, tiles_x, tiles_y;
    uint32_t num_pixels;
    uint32_t n;          // gaussians
    uint32_t num_nodes;  // wide nodes (0 if n == 0)
    // partition
    int rank, world;
    uint32_t num_tasks;  // tasks (one wave each) of this rank = its macro tiles x (256 / rays_per_task)
    uint32_t rays_per_task;   // 64: 8x8-pixel tasks (default); 32: 8x4; 16: 4x4 (under-filled ranks of a multi-GPU partition: egr_make_view)
    uint32_t task_shift;      // log2(tasks per 16x16 macro tile) = 2 / 3 / 4
    const uint32_t *task_macro; // [num_tasks >> task_shift] macro tile index of each group of tasks
    egr_gaussians g;
    egr_config cfg;
    egr_camera cam;
    egr_framebuffer fb;
    egr_metadata meta;
    egr_stats stats;
    const uint4 *wnodes;
    const uint32_t *gid_of_pos; // [n] gaussian id stored at sorted position p
    const uint32_t *pos_of_gid; // [n] inverse
    BvhFrame frame;
    const uint32_t *out_of_frame; // != 0: some box carries the -inf / +inf sentinel cells (see qslab_hit)
    float4 *inst_w;             // [n][4] test record: rows 0-2 = W (snapshot), row 3 = live quarter (k_live writes it per launch)
    float4 *inst_m;             // [n][4] backward record: rows 0-2 = (M row (snapshot), exp(scale_a) (live)), row 3 = raw quaternion (live)
    float *grad_rows;           // [n][32] gradient accumulation rows in record order (one 128-B line per gaussian)
    float4 *app;                // [n][2] live appearance (k_live writes it per launch)
    const float4 *bsph;         // [n] bounding sphere of the gaussian's ellipsoid (centre, squared radius; snapshot): the per-ray pre-test of primary tiles
    // per-launch scratch
    float *cand_keys;      // [slots][cand_cap][64]
    float2 *cand_vals;     // [slots][cand_cap][64]  (alpha, gaussian id bits)
    uint32_t *stack_spill; // [slots][EGR_GSTK * 64] pair-stack entries beyond the LDS part (one column per resident wave)
    uint32_t cand_cap;
    // candidate lists longer than cand_cap continue in an extension block (one per ray, EGR_EXT_BLOCK entries, bump-allocated per
    // launch): the reference's candidate pool is global, a single grazing ray may take thousands of entries
    float *ext_keys;
    float2 *ext_vals;
    uint32_t ext_blocks_cap;
    uint32_t num_slots;    // resident waves
    float4 *hit_arena;     // blocks of (1 + EGR_HIT_BLOCK_ROWS) rows x 64 lanes
    uint32_t hit_blocks_cap;
    uint32_t *task_last_block; // [NSTEPS][num_tasks]  last arena block of the task's step (or ~0u)
    uint32_t *task_cost;       // [num_tasks] grad launches: what the forward chain expects the task's backward to cost (hit rows / hits)
    uint32_t *bwd_order;       // [num_tasks] the order in which the backward chain takes this rank's tasks: costliest first inside every queue's chunk (k_order_backward)
    float *state;          // internal per-ray state, SoA by task-linear index (see trace.hip)
    uint32_t state_stride; // = padded number of task-linear rays
    uint32_t *control;     // device counters (see ControlWord)
    uint32_t task_begin, task_count; // this strand's slice of the rank's task order (see egr_trace_launch)
    uint32_t *queues;      // [strands][2 kernels][8 XCD heads] task queue heads; this view's strand starts at `queues`
    uint32_t num_strands;
    int grad_overwrite;    // k_grad_gather stores this launch's sums (per-launch buffer, egr_set_grad_overwrite) instead of adding them
    int team_help;         // forward chain: waves without tiles (or waiting for their own helpers) walk pairs their team mates offer (trace.hip: teams)
    const uint8_t *pixel_mask; // debug (egr_debug_set_pixel_mask): [H*W], a pixel with mask 0 is treated like a pixel outside the image; null = every pixel
    int cube_mode;         // exact-statistics launch (egr_set_exact_stats): the tree bounds instance CUBES, every overlap is counted
};

enum ControlWord : int {
    CW_ACCEPTED = 0,    // per-step 64-bit counters: accepted candidates = what the reference inserts into its forward list
    CW_HIT_BUMP = 128,  // arena block bump allocator: a returning atomic on the path of every eighth compositing batch - ON A CACHE LINE OF ITS OWN (words 128 .. 159;
                        // it used to be word 8, on the line every wave adds its twelve per-step counters to when it leaves the kernel)
    CW_STATUS = 9,
    CW_BUCKET_RECORDS = 7, // 64-B gradient records (wide adds) the backward chain sent to the gradient rows in this launch
    CW_EXT_BUMP = 160,     // candidate-list extension blocks handed out in this launch (its own line too: words 160 .. 191)
    CW_RAYS = 10,       // per-step 64-bit counters (two words each): rays[3], candidates[3], composited[3]
    CW_CAND = 16,
    CW_COMP = 22,
    CW_RESET_END = 28,  // words [0, CW_RESET_END) are zeroed by every launch
    CW_LIFE_RAYS = 28,  // lifetime totals (since egr_create / egr_reset_lifetime_counters), 64-bit
    CW_LIFE_LAUNCHES = 30,

    CW_DBG = 32,        // optional traversal statistics (EGR_TRAVERSAL_STATS builds): 8 x 64-bit
    CW_DBG2 = 48,       // per-phase s_memtime sums: [primary traversal, primary composite, bounce traversal, bounce composite]
    CW_DBG3 = 112,      // per forward step: min / max wave exit time (s_memrealtime)
    CW_COUNT = 192      // (words from CW_DBG on are zeroed by every launch's prologue)
};

struct KernelStamp {
    const char *name;
    hipEvent_t start, stop;
};

struct egr_context {
    int device = 0;
    int width = 0, height = 0;
    int64_t fwd_capacity = 0, bwd_capacity = 0;
    int rank = 0, world = 1;
    bool bound = false, have_gaussians = false, bvh_valid = false;
    bool live_fresh = false; // the last API call was an egr_update_bvh_ex(EGR_UPDATE_FUSE_LIVE) that wrote the live records; consumed (cleared) on entry of egr_raytrace, dropped by every other call that touches the context
    egr_gaussians g{};
    egr_config cfg{};
    egr_camera cam{};
    egr_framebuffer fb{};
    egr_metadata meta{};
    egr_stats stats{};
    // BVH
    uint32_t n_alloc = 0;      // capacity of per-gaussian buffers
    uint32_t n_built = 0;      // n the tree topology was built for
    uint4 *wnodes = nullptr;
    uint32_t num_wide = 0;
    uint32_t *wide_of = nullptr;          // build temporary: binary internal id -> wide node index
    std::vector<uint32_t> level_start;    // host: wide-node index range of each level of the wide tree
    uint32_t *pos_of_gid = nullptr; // gid_of_pos is vals_out (kept after the build)
    BvhFrame frame{0.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    float4 *inst_w = nullptr, *inst_m = nullptr, *app = nullptr, *bsph = nullptr;
    float *aabb = nullptr;             // [n][6] instance boxes (lo, hi)
    uint32_t *out_of_frame = nullptr;  // device flag written by the refit
    float *grad_rows = nullptr;        // [n_alloc][32] zero between launches (k_grad_gather empties what it reads)
    uint32_t max_depth = 0;
    // build temporaries
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint64_t *keys_in = nullptr, *keys_out = nullptr;
    uint32_t *vals_in = nullptr, *vals_out = nullptr;
    int32_t *k_left = nullptr, *k_right = nullptr, *k_parent = nullptr; // Karras arrays (index space: internal i, leaf n-1+j)
    uint32_t *k_first = nullptr, *k_last = nullptr;
    float *k_dp = nullptr;       // SAH collapse: [n_alloc][16] box + T(1..7) per binary internal node (bvh.hip: k_sah_bottom_up)
    uint32_t *k_flags = nullptr; // its arrival counters
    uint32_t *scratch_u32 = nullptr; // bounds (6), depth histogram, cursors
    // launch scratch
    float *cand_keys = nullptr;
    float2 *cand_vals = nullptr;
    uint32_t *stack_spill = nullptr;
    bool grad_overwrite = false;  // egr_set_grad_overwrite
    bool delta_pending = false;   // the per-launch buffer holds a grad launch the caller has not consumed yet (egr_grad_delta_consumed): the next grad launch ADDS
    const uint8_t *pixel_mask = nullptr; // egr_debug_set_pixel_mask (caller-owned device memory)
    bool exact_stats = false;     // egr_set_exact_stats: cube boxes + reference-defined candidate count (takes effect at the next update / rebuild)
    bool boxes_are_cubes = false; // what the current tree was refitted with
    int denoise_mode = 1;         // 1: a-trous stand-in (denoise.hip), 0: copy output_final
    float *ext_keys = nullptr;    // candidate-list extension blocks (see DeviceView)
    float2 *ext_vals = nullptr;
    uint32_t ext_blocks_cap = 0;
    float *denoise_tmp = nullptr; // two W*H*3 ping-pong images, allocated on first use
    // strands: the rank's tiles are cut into `strands` slices whose kernel sequences run on separate HIP streams, so one
    // slice's persistent-wave tail (few long tiles left) is filled by the other slice's next kernel
    int strands = 3;        // allocated (scratch, streams); 3 strands + the caller's stream = the 4 HW queues of the runtime
    int strands_active = 0; // used by the next launch (0 = all allocated)
    hipStream_t strand_stream[EGR_MAX_STRANDS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[EGR_MAX_STRANDS] = {};
    uint32_t *queues = nullptr;
    uint32_t cand_cap = 0, num_slots = 0;
    int team_waves_per_cu = 0; // resident waves per CU of the forward chain's team build (whole teams)
    float4 *hit_arena = nullptr;
    uint32_t hit_blocks_cap = 0;
    uint32_t *task_last_block = nullptr;
    uint32_t *task_cost = nullptr, *bwd_order = nullptr;
    float *state = nullptr;
    uint32_t state_stride = 0;
    uint32_t num_tasks_total = 0; // 8x8 wave tiles in the whole image (a 16x16 macro tile = 4 of them = 256 rays of ray state)
    int team_help = 1;            // egr_set_team_help / env EGR_TEAM_HELP: 1 (default) = waves without tiles help their team mates' walks, 0 = never, -1 = only for under-filled ranks of a partition (egr_team_help_on)
    int rays_per_task = 0;        // 0: automatic (64; 32 for a rank of a partition with fewer than two 8x8 tiles per wave slot); env EGR_RAYS_PER_TASK
    uint32_t *task_macro = nullptr; // device table of the current partition's tile order: one of task_orders[].table
    struct TaskOrder {              // tile orders built so far (a partitioned trainer flips between (rank, world) for training launches and
        int rank, world;            // (0, 1) for evaluation renders: egr_set_partition then only swaps a pointer, no sync, no upload)
        uint32_t *table;
    };
    std::vector<TaskOrder> task_orders;
    uint32_t *control = nullptr;
    uint32_t *control_host = nullptr; // pinned
    // timing
    bool timing = false;
    hipEvent_t ev_rt0 = nullptr, ev_rt1 = nullptr, ev_ub0 = nullptr, ev_ub1 = nullptr;
    bool have_rt = false, have_ub = false;
    std::vector<KernelStamp> stamps;
    size_t stamps_used = 0;
    std::string last_error;
    // device memory this context holds (egr_counters::device_bytes): every allocation goes through egr_dev_alloc / egr_dev_free
    std::unordered_map<void *, size_t> alloc_sizes;
    size_t device_bytes = 0;
};

inline void egr_dev_alloc_raw(egr_context *c, void **p, size_t bytes) {
    EGR_HIP(hipMalloc(p, bytes));
    c->alloc_sizes[*p] = bytes, c->device_bytes += bytes;
}
template <class T> void egr_dev_alloc(egr_context *c, T *&p, size_t count) { egr_dev_alloc_raw(c, (void **)&p, (count ? count : 1) * sizeof(T)); }
template <class T> void egr_dev_free(egr_context *c, T *&p) {
    if (p) {
        auto it = c->alloc_sizes.find((void *)p);
        if (it != c->alloc_sizes.end()) c->device_bytes -= it->second, c->alloc_sizes.erase(it);
        (void)hipFree((void *)p);
    }
    p = nullptr;
}

// bvh.hip
void egr_bvh_free(egr_context *c);
void egr_bvh_reserve(egr_context *c, uint32_t n);
void egr_bvh_rebuild(egr_context *c, hipStream_t s);
void egr_bvh_refit(egr_context *c, hipStream_t s, bool fuse_live = false);
int egr_bvh_check(egr_context *c, hipStream_t s, std::string &msg);
// trace.hip
void egr_trace_alloc(egr_context *c);
void egr_trace_free(egr_context *c);
void egr_trace_launch(egr_context *c, bool grads, bool live_fresh, hipStream_t s);
uint32_t egr_num_tasks_for_rank(const egr_context *c);
void egr_build_task_order(egr_context *c);
DeviceView egr_make_view(const egr_context *c);
bool egr_team_help_on(const egr_context *c); // the forward chain of the next launch runs as teams with help
// timing helpers (api.hip)
void egr_stamp_begin(egr_context *c, const char *name, hipStream_t s);
void egr_stamp_end(egr_context *c, hipStream_t s);
