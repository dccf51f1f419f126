// Wavefront path tracer for gfx950 (MI355X) and its extern "C" entry points (include/ptgpu.h).
//
// The per-pixel recursion of TraceSceneBatchJob::trace / color (src/trace.rs:117-234) becomes four stages:
//   raygen   camera samples -> rays                           (src/trace.rs:152-158, src/camera.rs:53-59)
//   extend   closest hit of every ray                         (Hitable::hit on SceneState.world, src/bvh.rs:74-99)
//   shade    emitted / scatter / light sampling, next ray     (src/trace.rs:205-226, src/material.rs)
//            + compaction: free slots are refilled by wave64 ballot + prefix count
//   resolve  per-pixel mean in sample order -> f32 RGBA image (src/trace.rs:166-182)
// One pass traces `paths in flight` paths to completion. Offline passes run the first three stages inside one kernel
// whose waves keep 64 paths in registers (kernels_pool.h); realtime frames, ray batches and the forms kept for
// comparison run them as launches over queues of path states in HBM (kernels_queue.h, kernels_extend.h,
// kernels_shade.h). This file holds the context, the choice of form (run_pass) and the C entry points.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h> // types and prototypes only: the library is loaded on first use, see rccl_api()

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <cstddef>
#include <string>
#include <type_traits>
#include <vector>

#include "accel.h"
#include "flatten.h"
#include "group_rows.h"
#include "pool_list.h"
#include "pt_device.h"

using namespace ptdev;

namespace {

#include "kernels_queue.h"
#include "kernels_extend.h"
#include "kernels_tree.h"
#include "kernels_shade.h"
#include "kernels_pool.h"
#include "kernels_post.h"

} // namespace

// ================================================================================================================
// Scratch of pt_intersect_static: the ray queue, the hit records and the queue's fill count of one batch of at most
// PT_INTERSECT_STATIC_MAX_RAYS rays, in a device array that exists from the moment the library is loaded, so that the
// entry point never allocates. Every GPU holds its own copy at its own address: `scratch` itself is only the host-side
// handle of the symbol, and the address kernels are given comes from hipGetSymbolAddress on the device in question
// (static_scratch_for, called by pt_create).
// ================================================================================================================
constexpr uint32_t kStaticMaxRays = PT_INTERSECT_STATIC_MAX_RAYS;
static_assert(kStaticMaxRays % 256 == 0, "whole workgroup chunks");
constexpr size_t kStaticQueueBytes = (size_t)(kStaticMaxRays / 64) * Queue::kTileBytes;
constexpr size_t kStaticHitTOffset = kStaticQueueBytes;
constexpr size_t kStaticHitIdOffset = kStaticHitTOffset + (size_t)kStaticMaxRays * sizeof(double);
constexpr size_t kStaticCountsOffset = kStaticHitIdOffset + (size_t)kStaticMaxRays * sizeof(int32_t);
constexpr size_t kStaticScratchBytes = kStaticCountsOffset + 256;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// Every allocation the library makes goes through these two, so that pt_device_alloc_count can say whether a call allocated.
static std::atomic<uint64_t> g_alloc_count{0};
template <typename T> static hipError_t counted_malloc(T **p, size_t bytes) {
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    return hipMalloc((void **)p, bytes);
}
static hipError_t counted_host_malloc(void **p, size_t bytes, unsigned int flags) {
    g_alloc_count.fetch_add(1, std::memory_order_relaxed);
    return hipHostMalloc(p, bytes, flags);
}

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {}; // device address of `scratch` on each GPU, once resolved

// The device address of `scratch` on `device`, which must be the current device. Resolved once per device.
hipError_t static_scratch_for(int device, unsigned char **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned char *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
} // namespace

// ================================================================================================================
// Context
// ================================================================================================================

struct pt_ctx {
    int device = 0;
    uint64_t seed = 0;
    uint32_t flags = 0;
    uint32_t capacity = 0; // paths in flight (rounded up to whole shards)
    QueueShape shape{};    // sharded layout of the render queues
    hipStream_t stream = nullptr;
    std::string error;
    int n_cu = 256;

    // scene
    bool have_scene = false, have_camera = false;
    FlatScene flat;
    SceneView view{};
    int tex = 0; // texture level of the uploaded scene: 0 constant colours only, 1 checkers too, 2 noise / image textures too
    void *d_ops = nullptr, *d_xforms = nullptr, *d_leaves = nullptr, *d_lights = nullptr, *d_materials = nullptr,
         *d_textures = nullptr, *d_texels = nullptr, *d_perlin = nullptr;
    CameraDev cam{};

    // path state
    void *d_pool = nullptr;   // one allocation carved into the radiance arrays below
    void *d_queues = nullptr; // one allocation carved into the queues and hit records (made when a form that needs them first runs)
    Queue q[2]{};
    double *d_l[3] = {nullptr, nullptr, nullptr};
    double *d_hit_t = nullptr;
    int32_t *d_hit_id = nullptr;
    uint32_t *d_counts = nullptr;
    uint32_t counts_clean_words = 0; // leading words of d_counts that the last operation on the stream left at zero
    unsigned long long *d_stats = nullptr; // [0] segments, [1] nan pixels
    TraversalCounters *d_trav = nullptr;

    // image
    uint32_t width = 0, height = 0;
    float *d_image = nullptr;
    bool image_external = false;
    float *d_image_owned = nullptr;
    uint32_t *d_frames = nullptr;
    double *d_acc[3] = {nullptr, nullptr, nullptr};
    size_t acc_capacity = 0;

    // profiling
    pt_stats stats{};
    std::vector<hipEvent_t> events;
    uint32_t profile_period = 8;
    uint64_t pass_counter = 0;
    std::vector<uint64_t> h_counts; // per bounce, summed over the shards
    std::vector<uint32_t> h_shard_counts;

    // launch shaping
    bool exact = false;            // PT_CFG_EXACT
    bool use_scan = false;         // wave-uniform extend for short op streams
    int scan_max_ops = 96;
    bool scan_list = false;        // ... over the stream with its BOX ops removed
    bool fuse = true;              // short box-free lists are traced and shaded in one kernel (off: PT_PASS_PER_BOUNCE)
    int list_max_leaves = 24;
    void *d_ops_list = nullptr;
    int32_t n_list_ops = 0;
    bool fuse_ok = false;          // the scene has a fused-kernel op list (d_ops_fused)
    uint32_t render_flags = 0;     // pt_render_args.flags of the render call in progress
    uint32_t persist_factor = 1;
    bool tree_pool = true;        // offline passes of tree scenes as one k_tree_pool launch
    uint32_t tree_tuning[4] = {16, 16, 32, 4}; // refill at / inner min / shade min / traversing min
    bool tree_refill_set = false;              // refill at came from pt_config (else big trees of plain spheres use 4)
    bool tree_inner_set = false;               // inner min came from pt_config (else big trees use 24, of plain spheres 28, realtime frames 4)
    bool tree_shade_set = false;               // shade min came from pt_config (else realtime frames use 48)
    void *d_pool_consts = nullptr;
    bool exact_pool = true;       // exact mode uses the path pool too (short streams)
    bool path_pool = true;        // offline fused passes as one k_path_pool launch
    uint32_t pool_rt_min = 0; // realtime frames of list scenes: path ids from which the pool is used (below: the queue form).
                              // A frame through the pool is one launch and wins at every size (320 x 180: 13 200 against 10 200 frames/s)
    uint32_t pool_refill = 4;     // free path slots at which a wave generates new camera samples
    uint32_t pool_grid = 1;       // workgroups per resident slot
    uint32_t pool_chunk = 512;    // path ids a wave takes from the pass's counter at a time
    bool accel_pool = true;      // SAH traversal with lane refill (k_extend_accel_pool) instead of a ray per thread
    uint32_t accel_refill = 24;  // free lanes at which a wave takes new rays
    uint32_t accel_pool_from = 0; // first bounce traced by the pool form
    bool accel_pool_from_set = false;
    uint32_t accel_grid = 2;     // workgroups per resident slot
    uint32_t accel_inner = 24;   // lanes on inner nodes below which a wave turns to its leaves
    uint32_t shards = 8;           // shards of the render queues
    bool use_accel = true;         // fast mode: SAH tree for scenes that are not a short list (PTGPU_ACCEL=0 disables)
    bool accel_ok = false;
    void *d_accel_nodes = nullptr, *d_accel_prims = nullptr, *d_accel_parent = nullptr, *d_accel_levels = nullptr;
    void *d_accel_nodes_q = nullptr; // the nodes in their 32-byte form (trees of kAccelQuantFrom nodes and more), else null
    void *d_accel_spheres = nullptr; // the primitives as compact DevSphere records in leaf order (scenes of plain spheres alone), else null
    bool compact_spheres = true;     // use them (PT_CFG_FAT_RECORDS turns them off, for A/B runs)
    bool quant_pool = true;          // the sphere tree pool reads the 32-byte nodes where they exist (PT_CFG_WIDE_NODES turns that off, for A/B runs)
    AccelGrid accel_grid_q{};
    int last_pass_records = 0;       // PT_RECORDS_* of the most recent pass
    int32_t accel_q_start[2] = {0, 0}; // where rays start in the 32-byte nodes: node (the root, or the leaf it holds) and the node pushed behind it
    int32_t n_accel_nodes = 0;
    std::vector<std::pair<int32_t, int32_t>> accel_levels; // refit order (deepest level first): index ranges of d_accel_levels, or node id ranges when that is null
    int accel_depth = 0;
    int32_t n_accel_prims = 0;
    void *d_ops_fused = nullptr;
    int32_t n_fused_ops = 0;
    std::vector<uint64_t> profile; // queue sizes per bounce of the last pass that was read back
    uint32_t profile_n = 0;        // its path count

    // Kept between calls and grown on demand, so that a loop of pt_intersect / pt_tonemap_device /
    // pt_display_transform calls allocates once: [0] the ray batch, [1] the hit batch or the post kernels' output.
    void *d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes[2] = {0, 0};
    hipEvent_t ev_batch[2] = {nullptr, nullptr}; // bracket the extend launches of a ray batch
    unsigned char *d_static = nullptr;           // this GPU's copy of `scratch` (pt_intersect_static)

    // membership in a group of contexts, one per GPU, that render one image together (pt_group_*)
    ncclComm_t comm = nullptr;
    int group_rank = 0, group_world = 1;

    // asynchronous gather (pt_group_gather_async): a second stream carries the transfers while the render stream traces
    // the next frame. Slot = gather number & 1.
    hipStream_t comm_stream = nullptr;
    void *d_gpack[2] = {nullptr, nullptr};      // this member's packed rows of the frame being gathered
    void *d_grecv[2] = {nullptr, nullptr};      // root: the other members' packed rows
    void *d_gathered[2] = {nullptr, nullptr};   // root: the assembled frames
    size_t gpack_bytes = 0, grecv_bytes = 0, gathered_bytes = 0;
    hipEvent_t ev_packed[2] = {nullptr, nullptr};   // render stream: the slot's rows are packed
    hipEvent_t ev_gathered[2] = {nullptr, nullptr}; // comm stream: the slot's gather is complete (staging free, frame assembled)
    hipEvent_t ev_consumed = nullptr;               // render stream: the last reader of an assembled frame is done
    bool consumed_pending = false;
    bool gather_used[2] = {false, false};
    uint64_t gather_seq = 0;
    int gathered_slot = -1; // root: the assembled frame pt_readback / pt_tonemap_device / pt_display_transform / pt_present_submit read; -1: the context's own image

    // presentation ring (pt_present_*): the host's staging memory, pinned, and one device snapshot per slot
    static constexpr uint32_t kPresentSlots = 4;
    unsigned char *present_host = nullptr;
    void *present_pinned = nullptr;                  // start of the pinned page range, null if pinning was refused
    bool present_owned = false;                      // the ring is the library's own hipHostMalloc allocation
    size_t present_slot_bytes = 0;
    uint32_t present_slots = 0;
    hipStream_t present_stream = nullptr;            // carries the copies to the host, beside the render stream
    void *d_present[kPresentSlots] = {};             // snapshot of the frame a slot shows
    size_t present_bytes[kPresentSlots] = {};
    hipEvent_t ev_snapshot[kPresentSlots] = {};      // render stream: the slot's snapshot is complete
    hipEvent_t ev_landed[kPresentSlots] = {};        // copy stream: the slot's bytes are in host memory
    bool present_pending[kPresentSlots] = {};
};

namespace {

int set_error(pt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                        \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            return set_error(ctx, e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP,                   \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
        }                                                                                                         \
    } while (0)

template <typename T> int upload(pt_ctx *ctx, void **dst, const T *src, size_t count) {
    if (*dst) {
        (void)hipFree(*dst);
        *dst = nullptr;
    }
    if (count == 0) return PT_OK;
    HIP_TRY(ctx, counted_malloc(dst, count * sizeof(T)));
    HIP_TRY(ctx, hipMemcpyAsync(*dst, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int grid_for(const pt_ctx *ctx, uint64_t n) {
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    uint64_t cap = (uint64_t)ctx->n_cu * 8; // 8 workgroups of 256 per CU = the 32-wave limit
    if (blocks < 1) blocks = 1;
    return (int)std::min(blocks, cap);
}

int alloc_path_state(pt_ctx *ctx) {
    // whole shards of whole workgroup chunks
    // measured: 64 shards cost 18 % (scattered streams), 8 cost 2 % and leave 8x atomic headroom
    const uint32_t shards = ctx->shards;
    uint32_t shift = 0;
    while ((1u << shift) < shards) ++shift;
    const uint32_t per_shard = ((ctx->capacity + kBlock * shards - 1) / (kBlock * shards)) * kBlock;
    ctx->capacity = per_shard * shards;
    ctx->shape.shards = shards;
    ctx->shape.shift = shift;
    ctx->shape.shard_cap = per_shard;
    const size_t n = ctx->capacity;
    // The per-path radiance (3 f64), which every pass form writes and the resolve reads: 24 bytes per path in flight. The
    // queues and hit records of the forms that keep paths in memory (204 bytes per path) are allocated when one of those
    // forms first runs (ensure_queues): the pool kernels, which keep their paths in registers, never touch them.
    const size_t bytes = n * 3 * 8 + 3 * 256; // every array starts on a 256-byte boundary
    HIP_TRY(ctx, counted_malloc(&ctx->d_pool, bytes));
    char *p = (char *)ctx->d_pool;
    auto take = [&](size_t sz) {
        char *r = p;
        p += (sz + 255) & ~(size_t)255;
        return (void *)r;
    };
    for (int k = 0; k < 3; ++k) ctx->d_l[k] = (double *)take(n * 8);
    HIP_TRY(ctx, counted_malloc((void **)&ctx->d_counts, (size_t)kMaxDepthSlots * kShards * sizeof(uint32_t)));
    HIP_TRY(ctx, counted_malloc((void **)&ctx->d_stats, 2 * sizeof(unsigned long long)));
    HIP_TRY(ctx, counted_malloc((void **)&ctx->d_trav, sizeof(TraversalCounters)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_trav, 0, sizeof(TraversalCounters), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// The two path queues (96 bytes per entry each) and the hit records (f64 t, i32 id) of the pass forms that keep their paths
// in memory, and of pt_intersect: allocated on first use, kept until the context is destroyed.
int ensure_queues(pt_ctx *ctx) {
    if (ctx->d_queues) return PT_OK;
    const size_t n = ctx->capacity;
    const size_t bytes = n * (2 * 96 + 8 + 4) + 4 * 256;
    HIP_TRY(ctx, counted_malloc(&ctx->d_queues, bytes));
    char *p = (char *)ctx->d_queues;
    auto take = [&](size_t sz) {
        char *r = p;
        p += (sz + 255) & ~(size_t)255;
        return (void *)r;
    };
    for (int k = 0; k < 2; ++k) ctx->q[k].base = (char *)take((n / 64) * (size_t)Queue::kTileBytes); // n is a multiple of kBlock
    ctx->d_hit_t = (double *)take(n * 8);
    ctx->d_hit_id = (int32_t *)take(n * 4);
    return PT_OK;
}

int ensure_image(pt_ctx *ctx, uint32_t w, uint32_t h) {
    if (ctx->image_external) {
        if (w != ctx->width || h != ctx->height)
            return set_error(ctx, PT_ERR_INVALID_ARG, "render size differs from the bound image");
    } else if (!ctx->d_image_owned || w != ctx->width || h != ctx->height) {
        if (ctx->d_image_owned) (void)hipFree(ctx->d_image_owned);
        ctx->d_image_owned = nullptr;
        HIP_TRY(ctx, counted_malloc((void **)&ctx->d_image_owned, (size_t)w * h * 4 * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_image_owned, 0, (size_t)w * h * 4 * sizeof(float), ctx->stream));
        ctx->d_image = ctx->d_image_owned;
        ctx->width = w;
        ctx->height = h;
        ctx->gathered_slot = -1; // an assembled frame of the old size is not this image
        if (ctx->d_frames) (void)hipFree(ctx->d_frames);
        ctx->d_frames = nullptr;
    }
    if (!ctx->d_frames) {
        HIP_TRY(ctx, counted_malloc((void **)&ctx->d_frames, (size_t)w * h * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_frames, 0, (size_t)w * h * sizeof(uint32_t), ctx->stream));
    }
    return PT_OK;
}

int ensure_acc(pt_ctx *ctx, size_t pixels) {
    if (pixels <= ctx->acc_capacity) return PT_OK;
    for (int k = 0; k < 3; ++k) {
        if (ctx->d_acc[k]) (void)hipFree(ctx->d_acc[k]);
        ctx->d_acc[k] = nullptr;
    }
    ctx->acc_capacity = 0;
    for (int k = 0; k < 3; ++k) HIP_TRY(ctx, counted_malloc((void **)&ctx->d_acc[k], pixels * sizeof(double)));
    ctx->acc_capacity = pixels;
    return PT_OK;
}

int ensure_stage(pt_ctx *ctx, int k, size_t bytes) {
    if (bytes <= ctx->stage_bytes[k]) return PT_OK;
    if (ctx->d_stage[k]) (void)hipFree(ctx->d_stage[k]);
    ctx->d_stage[k] = nullptr;
    ctx->stage_bytes[k] = 0;
    HIP_TRY(ctx, counted_malloc(&ctx->d_stage[k], bytes));
    ctx->stage_bytes[k] = bytes;
    return PT_OK;
}

// True when [p, p + bytes) lies inside one allocation of device memory on `device` (a kernel handed anything else
// would fault the GPU instead of returning an error).
bool device_range_ok(int device, const void *p, size_t bytes) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // a plain host pointer: not an error of the stream
        return false;
    }
    if (attr.type != hipMemoryTypeDevice || attr.device != device) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const char *b = (const char *)base, *q = (const char *)p;
    return q >= b && bytes <= size && (size_t)(q - b) <= size - bytes;
}

// The image the reading entry points (pt_readback, pt_tonemap_device, pt_display_transform, pt_present_submit) take:
// the context's own, or on a group's root after pt_group_gather_async the latest assembled frame, which the render
// stream first waits for. shown_done marks the end of such a read, so that the slot is not reassembled under it.
int shown_image(pt_ctx *ctx, const float **image) {
    *image = ctx->d_image;
    if (ctx->gathered_slot >= 0) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[ctx->gathered_slot], 0));
        *image = (const float *)ctx->d_gathered[ctx->gathered_slot];
    }
    return PT_OK;
}
int shown_done(pt_ctx *ctx) {
    if (ctx->gathered_slot >= 0) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_consumed, ctx->stream));
        ctx->consumed_pending = true;
    }
    return PT_OK;
}

D3 hd3(const double p[3]) { return D3{p[0], p[1], p[2]}; }
D3 hsub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 hmul(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
D3 hcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x}; }
D3 hunit(D3 a) {
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return D3{a.x / l, a.y / l, a.z / l};
}

template <bool X, bool FRAMES, bool MEDIA>
void launch_extend_xfm(pt_ctx *ctx, int grid, const Queue &q, const ExtendArgs &a) {
    if (MEDIA && ctx->flat.general_boundaries) {
        // a medium whose boundary is a list or a tree: the per-lane walk of the reference's own tree is the one form that
        // carries the boundary walk (no recipe has such a medium; the forms below keep their registers for those that do)
        hipLaunchKernelGGL((k_extend<X, FRAMES, 2>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view, a);
        return;
    }
    if (!X && ctx->accel_ok && a.counters != nullptr && (ctx->flags & PT_CFG_COUNT_ACCEL)) {
        // diagnostic: the ray-per-thread form with its node and primitive counters on
        hipLaunchKernelGGL((k_extend_accel<MEDIA, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                           (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a);
    } else if (!X && ctx->accel_ok && a.counters == nullptr && ctx->accel_pool && !(ctx->accel_pool_from > a.depth && a.key_mode == 0)) {
        grid = std::min(grid, ctx->n_cu * (MEDIA ? 4 : 5) * (int)ctx->accel_grid); // what is resident at once
        if (ctx->d_accel_nodes_q && ctx->n_accel_nodes >= kAccelQuantFrom) // (a smaller tree's grid may leave out the root's boxes)
            hipLaunchKernelGGL((k_extend_accel_pool<MEDIA, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                               (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a,
                               ctx->accel_refill | (ctx->accel_inner << 8), (const AccelNodeQ *)ctx->d_accel_nodes_q, ctx->accel_grid_q);
        else
            hipLaunchKernelGGL((k_extend_accel_pool<MEDIA, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                               (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a,
                               ctx->accel_refill | (ctx->accel_inner << 8), (const AccelNodeQ *)nullptr, AccelGrid{});
    } else if (!X && ctx->accel_ok && a.counters == nullptr) {
        hipLaunchKernelGGL((k_extend_accel<MEDIA>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                           (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a);
    } else if (!X && !FRAMES && ctx->use_scan && ctx->scan_list && ctx->fuse_ok && a.counters == nullptr) {
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
        hipLaunchKernelGGL((k_extend_list<MEDIA>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, v, a);
    } else if (ctx->use_scan && ctx->scan_list) {
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_list;
        v.n_traversal_ops = ctx->n_list_ops;
        hipLaunchKernelGGL((k_extend_scan<X, FRAMES, MEDIA, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, v, a);
    } else if (ctx->use_scan)
        hipLaunchKernelGGL((k_extend_scan<X, FRAMES, MEDIA, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, q,
                           ctx->view, a);
    else
        hipLaunchKernelGGL((k_extend<X, FRAMES, MEDIA ? 1 : 0>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view, a);
}
template <bool X> void launch_extend_x(pt_ctx *ctx, int grid, const Queue &q, const ExtendArgs &a) {
    const bool frames = ctx->view.has_frames != 0, media = ctx->view.has_medium != 0;
    if (frames && media) launch_extend_xfm<X, true, true>(ctx, grid, q, a);
    else if (frames) launch_extend_xfm<X, true, false>(ctx, grid, q, a);
    else if (media) launch_extend_xfm<X, false, true>(ctx, grid, q, a);
    else launch_extend_xfm<X, false, false>(ctx, grid, q, a);
}
void launch_extend(pt_ctx *ctx, int grid, const Queue &q, const ExtendArgs &a) {
    if (ctx->exact) launch_extend_x<true>(ctx, grid, q, a);
    else launch_extend_x<false>(ctx, grid, q, a);
}
// Calls f with the context's texture level as a compile-time constant (std::integral_constant<int, 0 | 1 | 2>).
template <class F> void with_tex(int tex, F &&f) {
    if (tex == 0) f(std::integral_constant<int, 0>{});
    else if (tex == 1) f(std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 2>{});
}

void launch_shade(pt_ctx *ctx, int grid, const Queue &qi, const Queue &qo, uint32_t d, uint32_t max_depth) {
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, qi, qo, ctx->shape,
                           ctx->d_counts + (size_t)d * kShards, ctx->d_counts + (size_t)(d + 1) * kShards, ctx->view,
                           ctx->d_hit_t, ctx->d_hit_id, ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], d, max_depth,
                           ctx->render_flags);
    };
    with_tex(ctx->tex, [&](auto T) {
        constexpr int kTex = decltype(T)::value;
        if (ctx->exact) launch(k_shade<true, kTex>);
        else launch(k_shade<false, kTex>);
    });
}

// Workgroups for bounce d of a pass over n paths. Kernels grid-stride over the device-side queue size, so any
// grid is correct; the size only matters for speed. Deep bounces hold few rays, and launching the full grid for
// them costs more than the work itself, so the grid follows the queue sizes seen in the last pass read back.
int grid_for_bounce(const pt_ctx *ctx, uint32_t n, uint32_t d) {
    if (ctx->profile.empty() || ctx->profile_n == 0 || d >= ctx->profile.size()) return grid_for(ctx, n);
    const double scale = (double)n / (double)ctx->profile_n;
    const uint64_t guess = (uint64_t)(1.25 * scale * (double)ctx->profile[d]) + 2048;
    return grid_for(ctx, std::min<uint64_t>(guess, n));
}

// One wavefront pass over `pp.cpix * pp.s_count` paths. `read_counts` copies the per-bounce queue sizes back
// (one stream synchronisation) to shape the grids of the following passes.
int run_pass(pt_ctx *ctx, const PassParams &pp, bool read_counts) {
    const uint32_t n = pp.cpix * pp.s_count;
    const bool realtime = pp.mode == PT_MODE_REALTIME;
    const bool profile = (ctx->flags & PT_CFG_PROFILE_STAGES) && (ctx->pass_counter % ctx->profile_period == 0);
    const bool count = (ctx->flags & PT_CFG_COUNT_TRAVERSAL) != 0;
    const bool fused = ctx->fuse && ctx->scan_list && ctx->fuse_ok && !ctx->exact && !count && !ctx->flat.general_boundaries;
    ctx->pass_counter++;
    hipStream_t st = ctx->stream;
    const uint32_t iters = pp.max_depth + 1;
    size_t ev = 0;
    if (profile) {
        const size_t need = 2 * (2 + 2 * (size_t)iters);
        while (ctx->events.size() < need) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->events.push_back(e);
        }
    }
    auto mark = [&]() -> hipError_t { return profile ? hipEventRecord(ctx->events[ev++], st) : hipSuccess; };

    const int g_all = grid_for(ctx, n);
    // One launch per pass wherever a form for it exists. Offline passes keep their paths in registers (the pool
    // kernels). Realtime frames too (the waves evaluate the pixel skip before they make samples, and a frame of one
    // sample per pixel is one launch in all: its paths blend their own pixels), at every size: against the persistent
    // queue form below, 320 x 180 +29 %, 640 x 360 +15 %, 1280 x 720 +51 %, with media +40 to +120 %. (The queue form
    // itself measured +53 % at 2^20 paths, +13 % at 2^22, -3 % at 2^24 against a launch per bounce.)
    const bool pool = fused && ctx->path_pool && (!realtime || n >= ctx->pool_rt_min);
    // exact mode, short streams: the same pool around the wave-uniform walk of the reference's own tree
    const bool xpool = ctx->exact && ctx->exact_pool && ctx->path_pool && ctx->use_scan && !count && !realtime &&
                       ctx->view.n_leaves <= kLdsLeaves && !ctx->flat.general_boundaries;
    const bool tree = !fused && ctx->tree_pool && ctx->accel_ok && !ctx->exact && !count &&
                      !ctx->flat.general_boundaries; // offline passes and realtime frames
    const bool persistent = fused && !pool;
    // the persistent kernel deals contiguous runs of tiles to its waves: one shard, entry = path id
    const QueueShape gen_shape = persistent ? QueueShape{1u, 0u, ctx->capacity} : ctx->shape;
    // the pass's counters start at zero: cleared by the launch that sends a pool kernel its constants, where there is one
    // (a realtime frame is four operations on the stream, and a fifth of a millisecond notices each of them)
    const uint32_t n_counts = (iters + 1) * kShards;
    if (!(pool || tree || xpool)) {
        if (int rc = ensure_queues(ctx); rc != PT_OK) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_counts, 0, (size_t)n_counts * sizeof(uint32_t), st));
    }
    // A realtime frame through a pool kernel gets its constants with the launch packet (the kernel's first argument) and
    // finds its counters cleared by the previous frame's resolve: two operations on the stream instead of three.
    const bool counts_read_back = profile || (read_counts && !(persistent || pool || tree || xpool));
    // One sample per pixel in one pass (every frame of the reference's realtime loop): each path owns its pixel and blends
    // it into the image itself when it ends (blend_pixel), the frame's chunks are dealt by wave index, its counters are not
    // touched, and the frame is ONE operation on the stream. Profiled frames keep the resolve launch and its events.
    const bool resolve_inline = realtime && (pool || tree) && !counts_read_back && pp.spp == 1 && pp.s_count == 1 &&
                                pp.first_pass && pp.last_pass && ctx->d_frames != nullptr;
    const bool consts_by_launch = realtime && (pool || tree) && (resolve_inline || ctx->counts_clean_words >= n_counts);
    const uint32_t resolve_clears = (realtime && (pool || tree) && !counts_read_back && !resolve_inline) ? n_counts : 0u;
    const uint32_t counts_clean_before = ctx->counts_clean_words;
    ctx->counts_clean_words = 0; // until this pass's resolve has run
    HIP_TRY(ctx, mark());
    const PoolConsts pool_consts{ctx->cam, pp, ctx->accel_grid_q, ctx->accel_q_start[0], ctx->accel_q_start[1],
                                 ctx->d_image, ctx->d_frames, ctx->d_stats, resolve_inline ? 1u : 0u, 0u};
    auto send_pool_consts = [&]() -> hipError_t {
        if (consts_by_launch) return hipSuccess;
        if (!ctx->d_pool_consts)
            if (hipError_t e = counted_malloc(&ctx->d_pool_consts, sizeof(PoolConsts)); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_set_pool_consts, dim3(1), dim3(64), 0, st, pool_consts, (PoolConsts *)ctx->d_pool_consts,
                           ctx->d_counts, n_counts);
        return hipGetLastError();
    };
    if (pool || tree || xpool) {
        // no ray generation launch: the pool kernels make their own camera samples
    } else if (realtime)
        hipLaunchKernelGGL(k_raygen<true>, dim3(g_all), dim3(kBlock), 0, st, ctx->q[0], gen_shape,
                           ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->cam, pp, ctx->d_counts);
    else
        hipLaunchKernelGGL(k_raygen<false>, dim3(g_all), dim3(kBlock), 0, st, ctx->q[0], gen_shape,
                           ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->cam, pp, ctx->d_counts);
    HIP_TRY(ctx, mark());
    if (persistent) {
        // as many workgroups as stay resident (PT_SHADE_WAVES per SIMD = that many 4-wave workgroups per CU), or
        // fewer when the pass has fewer tiles than that many waves
        const uint32_t tiles = (n + 63u) / 64u;
        const uint32_t resident = (uint32_t)ctx->n_cu * PT_SHADE_WAVES * ctx->persist_factor;
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(resident, (tiles + kWavesPerBlock - 1) / kWavesPerBlock));
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        const bool media = ctx->view.has_medium != 0;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, ctx->q[0], ctx->q[1], ctx->d_counts, v,
                               ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], pp.max_depth, ctx->render_flags);
        };
        with_tex(ctx->tex, [&](auto T) {
            constexpr int kTex = decltype(T)::value;
            if (media) launch(k_trace_persistent<kTex, 2>);
            else launch(k_trace_persistent<kTex, 1>);
        });
        HIP_TRY(ctx, mark());
    }
    if (pool || xpool) {
        const uint32_t tiles = (n + 63u) / 64u;
        const bool media = ctx->view.has_medium != 0;
        const bool frames = ctx->view.has_frames != 0;
        // the kernel's FUSE parameter: 1 / 2 without / with media; exact mode on a stream without ENTER / EXIT ops: 3 / 4
        const int fuse_id = (xpool && !frames) ? (media ? 4 : 3) : (media ? 2 : 1);
        const uint32_t resident = (uint32_t)ctx->n_cu * pool_waves(xpool, fuse_id) * ctx->pool_grid;
        uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(resident, (tiles + kWavesPerBlock - 1) / kWavesPerBlock));
        if (realtime) {
            // A frame takes as long as its longest wave's chain of rounds, and a round is the shorter the fewer waves share
            // a SIMD. A frame that would leave the waves of a full grid a handful of paths each goes to fewer, fuller
            // waves: n / 1200 path ids per wave, between 64 and 512 (a fifth of them survive the pixel skip). 1080p keeps
            // its 506 ids per wave; Cornell box 1280 x 720: 7 140 -> 7 820 frames/s, 800 x 450: 9 090 -> 10 200,
            // 640 x 360: 8 000 -> 11 000, 480 x 270: 11 000 -> 13 100; beyond 512 ids per wave every size loses.
            const uint32_t ids = std::min(512u, std::max(64u, (n / 1200u + 63u) & ~63u));
            grid = std::max<uint32_t>(1u, std::min<uint32_t>(grid, (n + ids * kWavesPerBlock - 1) / (ids * kWavesPerBlock)));
        }
        // a small pass is cut into at least four chunks per wave so that the waves finish together
        // (a realtime frame's ids are mostly skipped pixels: one chunk per wave, or the chunk counter, one address for the
        // whole chip, is what the frame waits for)
        const uint32_t waves = grid * kWavesPerBlock;
        const uint32_t fair = ((realtime ? (n + waves - 1u) / waves : n / (waves * 4u)) + 63u) & ~63u;
        const uint32_t chunk = std::max(64u, realtime ? fair : std::min(ctx->pool_chunk, fair));
        // realtime: when every wave of the launch has a chunk of its own, the chunks are dealt by wave index
        const uint32_t static_bit = (realtime && (uint64_t)chunk * grid * kWavesPerBlock >= n) ? 0x80000000u : 0u;
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, send_pool_consts());
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, pool_consts, v, (const PoolConsts *)ctx->d_pool_consts, n,
                               pp.max_depth, ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags,
                               ctx->pool_refill | static_bit, chunk);
        };
        if (xpool) v = ctx->view; // the reference's own tree
        with_tex(ctx->tex, [&](auto T) {
            constexpr int kTex = decltype(T)::value;
            if (xpool) {
                if (media) {
                    if (frames) launch(k_path_pool<kTex, 2, true>);
                    else launch(k_path_pool<kTex, 4, true>);
                } else if (frames) launch(k_path_pool<kTex, 1, true>);
                else launch(k_path_pool<kTex, 3, true>);
            } else if (media) {
                if (realtime) launch(k_path_pool<kTex, 2, false, true>);
                else launch(k_path_pool<kTex, 2>);
            }
        });
        // fast mode without media: compiled in pool_list.hip, under a scheduling strategy of its own
        if (!xpool && !media)
            launch_path_pool_list(ctx->tex, realtime, grid, st, v, &pool_consts, ctx->d_pool_consts, n, pp.max_depth, ctx->d_l[0], ctx->d_l[1],
                                  ctx->d_l[2], ctx->d_counts, ctx->render_flags, ctx->pool_refill | static_bit, chunk);
        HIP_TRY(ctx, mark());
    }
    ctx->last_pass_records = 0;
    if (tree) {
        const uint32_t tiles = (n + 63u) / 64u;
        const bool spheres = ctx->d_accel_spheres && ctx->compact_spheres && ctx->tex <= 1 && !ctx->view.has_medium;
        const uint32_t resident = (uint32_t)ctx->n_cu * (ctx->tex == 2 ? PT_TREE_UV_WAVES : spheres ? PT_SPH_WAVES : PT_SHADE_WAVES) * ctx->pool_grid;
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(resident, (tiles + kWavesPerBlock - 1) / kWavesPerBlock));
        const uint32_t waves = grid * kWavesPerBlock;
        const uint32_t fair = ((realtime ? (n + waves - 1u) / waves : n / (waves * 4u)) + 63u) & ~63u; // as for the path pool
        // (fewer, fuller waves for small frames, as the path pool takes them, cost this kernel 2 to 40 %)
        uint32_t chunk = std::max(64u, realtime ? fair : std::min(ctx->pool_chunk, fair));
        if (realtime && (uint64_t)chunk * grid * kWavesPerBlock >= n) chunk |= 0x80000000u; // chunks dealt by wave index
        HIP_TRY(ctx, send_pool_consts());
        // big trees (a million spheres: a node fetch is an L2 miss more often than not) turn to their leaves earlier: the
        // lanes left on inner nodes would each wait a full memory round trip per step (one million spheres +3 %; the
        // small trees lose 4 % with it)
        // A realtime frame gives a wave some hundred paths in all, and its lanes thin out soon: the few that are still on
        // inner nodes must keep stepping, and shading waits for a fuller batch (1080p frames: final scene 616 -> 668 per
        // second, RTiOW 2 307 -> 2 528).
        // The variant for plain spheres on a big tree (32-byte nodes, a vertex of a few hundred instructions) takes new paths
        // as soon as four lanes are free and leaves its inner nodes at 28 lanes (one million spheres: 710 -> 773 Mpaths/s
        // against 16 / 24; refill at 1 .. 6 within 0.5 %, inner 20 / 32 / 40: 742 / 763 / 699).
        // A small tree of compact sphere records (random_scene): a vertex is cheap next to a round of the traversal, so
        // shading waits for 44 ready lanes, the lanes on inner nodes step on down to 8 of them, and new paths come in
        // batches of 24 (7 000 -> 7 450 Mpaths/s against 32 / 16 / 16; each of the three within 1 % of its neighbours).
        const bool big = ctx->n_accel_nodes >= kAccelQuantFrom;
        const bool small_spheres = spheres && !big && !realtime;
        const uint32_t inner_min = ctx->tree_inner_set ? ctx->tree_tuning[1]
                                   : realtime ? 4u : (big ? (spheres ? 28u : 24u) : small_spheres ? 8u : ctx->tree_tuning[1]);
        const uint32_t shade_min = ctx->tree_shade_set ? ctx->tree_tuning[2] : (realtime ? 48u : small_spheres ? 44u : ctx->tree_tuning[2]);
        const uint32_t refill_at = ctx->tree_refill_set ? ctx->tree_tuning[0]
                                   : ((big && spheres && !realtime) ? 4u : small_spheres ? 24u : ctx->tree_tuning[0]);
        const uint32_t tuning = refill_at | (inner_min << 8) | (shade_min << 16) | (ctx->tree_tuning[3] << 24);
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        const bool media = ctx->view.has_medium != 0;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, pool_consts, ctx->view, (const AccelNode *)ctx->d_accel_nodes,
                               (const DevOp *)ctx->d_accel_prims, (const PoolConsts *)ctx->d_pool_consts, n, pp.max_depth,
                               ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags, tuning, chunk);
        };
        // (the 32-byte nodes the extend kernel reads on big trees cost this kernel 12 more registers of ray state: at
        // 128 VGPRs that is 80 B more scratch per lane, and the million-sphere render ran 588 -> 503 Mpaths/s)
        if (spheres) {
            // plain spheres alone: the compact records stand in for the primitive ops (and the leaf table)
            auto launch_sph = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, pool_consts, ctx->view, (const AccelNode *)ctx->d_accel_nodes,
                                   (const DevOp *)ctx->d_accel_spheres, (const PoolConsts *)ctx->d_pool_consts, n, pp.max_depth,
                                   ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags, tuning, chunk);
            };
            auto launch_sph_q = [&](auto kernel) { // big trees: the 32-byte nodes
                hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, pool_consts, ctx->view, (const AccelNode *)ctx->d_accel_nodes_q,
                                   (const DevOp *)ctx->d_accel_spheres, (const PoolConsts *)ctx->d_pool_consts, n, pp.max_depth,
                                   ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags, tuning, chunk);
            };
            const bool quant = ctx->d_accel_nodes_q != nullptr && ctx->quant_pool;
            ctx->last_pass_records = PT_RECORDS_COMPACT_SPHERES | (quant ? PT_RECORDS_NODES32 : 0);
            if (ctx->tex == 0) {
                if (quant) {
                    if (realtime) launch_sph_q(k_tree_pool<0, false, true, true, true>);
                    else launch_sph_q(k_tree_pool<0, false, false, true, true>);
                } else if (realtime) launch_sph(k_tree_pool<0, false, true, true>);
                else launch_sph(k_tree_pool<0, false, false, true>);
            } else if (quant) {
                if (realtime) launch_sph_q(k_tree_pool<1, false, true, true, true>);
                else launch_sph_q(k_tree_pool<1, false, false, true, true>);
            } else if (realtime) launch_sph(k_tree_pool<1, false, true, true>);
            else launch_sph(k_tree_pool<1, false, false, true>);
        } else
        with_tex(ctx->tex, [&](auto T) {
            constexpr int kTex = decltype(T)::value;
            if (realtime) {
                if (media) launch(k_tree_pool<kTex, true, true>);
                else launch(k_tree_pool<kTex, false, true>);
            } else if (media) launch(k_tree_pool<kTex, true, false>);
            else launch(k_tree_pool<kTex, false, false>);
        });
        HIP_TRY(ctx, mark());
    }
    const bool one_launch = persistent || pool || tree || xpool;
    for (uint32_t d = one_launch ? iters : 0; d < iters; ++d) {
        const Queue &qi = ctx->q[d & 1];
        const Queue &qo = ctx->q[(d + 1) & 1];
        const int g_d = grid_for_bounce(ctx, n, d);
        ExtendArgs ea;
        ea.counts = ctx->d_counts + (size_t)d * kShards;
        ea.sh = ctx->shape;
        ea.t_min = 0.001; // src/trace.rs:205
        ea.t_max = kF64Max;
        ea.medium_seed = 0;
        ea.key_mode = 0;
        ea.depth = d;
        ea.hit_t = ctx->d_hit_t;
        ea.hit_id = ctx->d_hit_id;
        ea.counters = count ? ctx->d_trav : nullptr;
        HIP_TRY(ctx, mark());
        launch_extend(ctx, g_d, qi, ea);
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        launch_shade(ctx, g_d, qi, qo, d, pp.max_depth);
        HIP_TRY(ctx, mark());
    }
    HIP_TRY(ctx, mark());
    if (!resolve_inline)
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, pp.cpix)), dim3(kBlock), 0, st, ctx->d_l[0], ctx->d_l[1],
                           ctx->d_l[2], ctx->d_acc[0], ctx->d_acc[1], ctx->d_acc[2], ctx->d_image, ctx->d_frames, pp,
                           ctx->d_counts, ctx->d_stats, resolve_clears);
    HIP_TRY(ctx, mark());
    HIP_TRY(ctx, hipGetLastError());
    ctx->counts_clean_words = resolve_inline ? counts_clean_before : resolve_clears; // (such a frame leaves the counters alone)
    ctx->stats.paths += n;
    ctx->stats.launches_raygen += (pool || tree || xpool) ? 0 : 1;
    ctx->stats.launches_extend += (fused || tree || xpool) ? 0 : iters;
    ctx->stats.launches_shade += one_launch ? 1 : iters;
    ctx->stats.launches_resolve += resolve_inline ? 0 : 1;
    ctx->stats.last_pass_form = tree ? PT_PASS_TREE_POOL : (pool || xpool) ? PT_PASS_PATH_POOL : persistent ? PT_PASS_PERSISTENT
                                : PT_PASS_PER_BOUNCE;
    if (profile || (read_counts && !one_launch)) { // the one-launch forms size no grids from the counts
        ctx->h_shard_counts.resize((size_t)(iters + 1) * kShards);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_shard_counts.data(), ctx->d_counts,
                                    ctx->h_shard_counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        ctx->h_counts.assign(iters + 1, 0);
        for (uint32_t d = 0; d <= iters; ++d)
            for (uint32_t k = 0; k < kShards; ++k) ctx->h_counts[d] += ctx->h_shard_counts[(size_t)d * kShards + k];
        ctx->profile = ctx->h_counts;
        ctx->profile_n = n;
    }
    if (profile) {
        auto elapsed = [&](size_t a) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ctx->events[a], ctx->events[a + 1]);
            return (double)ms;
        };
        size_t k = 0;
        ctx->stats.ms_raygen += elapsed(k);
        k += 2;
        for (uint32_t d = 0; d < (one_launch ? 1u : iters); ++d) {
            ctx->stats.ms_extend += elapsed(k);
            k += 2;
            ctx->stats.ms_shade += elapsed(k);
            k += 2;
        }
        // segments of the launches timed as extend / shade
        for (uint32_t d = 0; d < iters; ++d) ctx->stats.profiled_segments += ctx->h_counts[d];
        ctx->stats.ms_resolve += elapsed(k);
        ctx->stats.profiled_paths += n;
        ctx->stats.profiled_passes += 1;
        ctx->stats.profiled_bounce_launches += one_launch ? 1 : iters;
    }
    return PT_OK;
}

// RCCL is loaded when a group is first formed, not when this library is: a host that never forms a group does not
// pay for mapping it, and a process that already holds a copy (PyTorch ships its own under the same soname) keeps
// using that one instead of getting a second.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
    decltype(&ncclSend) Send;
    decltype(&ncclRecv) Recv;
    decltype(&ncclGetErrorString) GetErrorString;
};
const RcclApi *rccl_api() {
    static const RcclApi *api = []() -> const RcclApi * {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return nullptr;
        static RcclApi a;
        bool ok = true;
        auto sym = [&](auto &fn, const char *name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
            ok = ok && fn != nullptr;
        };
        sym(a.GetUniqueId, "ncclGetUniqueId");
        sym(a.CommInitRank, "ncclCommInitRank");
        sym(a.CommDestroy, "ncclCommDestroy");
        sym(a.GroupStart, "ncclGroupStart");
        sym(a.GroupEnd, "ncclGroupEnd");
        sym(a.Send, "ncclSend");
        sym(a.Recv, "ncclRecv");
        sym(a.GetErrorString, "ncclGetErrorString");
        return ok ? &a : nullptr;
    }();
    return api;
}
// pt_tuning -> context. A zero field keeps the library default. Returns false on a value outside its range.
bool apply_tuning(pt_ctx *ctx, const pt_tuning &t) {
    auto in = [](uint32_t v, uint32_t lo, uint32_t hi) { return v >= lo && v <= hi; };
    switch (t.pass_form) {
    case PT_PASS_NONE: break;
    case PT_PASS_PER_BOUNCE: // every stage its own launch: no pools, no fused trace + shade
        ctx->fuse = ctx->path_pool = ctx->tree_pool = ctx->exact_pool = false;
        break;
    case PT_PASS_PERSISTENT: // list scenes through the persistent queue form, offline passes included
        ctx->path_pool = false;
        break;
    case PT_PASS_PATH_POOL: ctx->pool_rt_min = 0; break; // realtime frames of any size through the pool too
    case PT_PASS_TREE_POOL: break; // what offline passes use anyway
    default: return false;
    }
    if (t.pool_refill) { if (!in(t.pool_refill, 1, 64)) return false; ctx->pool_refill = t.pool_refill; }
    if (t.pool_chunk) { if (!in(t.pool_chunk, 64, 1u << 20)) return false; ctx->pool_chunk = t.pool_chunk; }
    if (t.pool_grid) { if (!in(t.pool_grid, 1, 256)) return false; ctx->pool_grid = t.pool_grid; }
    // a refill threshold above 64 would never be met: the kernel could not end
    if (t.tree_refill) { if (!in(t.tree_refill, 1, 64)) return false; ctx->tree_tuning[0] = t.tree_refill; ctx->tree_refill_set = true; }
    if (t.tree_inner) { if (!in(t.tree_inner, 1, 64)) return false; ctx->tree_tuning[1] = t.tree_inner; ctx->tree_inner_set = true; }
    if (t.tree_shade) { if (!in(t.tree_shade, 1, 64)) return false; ctx->tree_tuning[2] = t.tree_shade; ctx->tree_shade_set = true; }
    if (t.tree_trav_plus1) { if (!in(t.tree_trav_plus1, 1, 65)) return false; ctx->tree_tuning[3] = t.tree_trav_plus1 - 1; }
    if (t.accel_form) {
        if (!in(t.accel_form, 1, 2)) return false;
        ctx->accel_pool = t.accel_form == 2;
        ctx->accel_pool_from = 0;
        ctx->accel_pool_from_set = true;
    }
    if (t.accel_refill) { if (!in(t.accel_refill, 1, 64)) return false; ctx->accel_refill = t.accel_refill; }
    if (t.accel_inner) { if (!in(t.accel_inner, 1, 65)) return false; ctx->accel_inner = t.accel_inner; }
    if (t.accel_grid) { if (!in(t.accel_grid, 1, 8)) return false; ctx->accel_grid = t.accel_grid; }
    if (t.shards) { if (!in(t.shards, 1, kShards) || (t.shards & (t.shards - 1))) return false; ctx->shards = t.shards; }
    if (t.scan_max_ops_plus1) ctx->scan_max_ops = (int)std::min<uint32_t>(t.scan_max_ops_plus1 - 1, 1u << 30);
    if (t.list_max_leaves_plus1) ctx->list_max_leaves = (int)std::min<uint32_t>(t.list_max_leaves_plus1 - 1, 1u << 30);
    if (t.profile_period) ctx->profile_period = t.profile_period;
    return true;
}

// PTGPU_TUNING="field=value,field=value": overrides fields of pt_tuning by name, for A/B runs of a host that is not
// rebuilt. Unknown names and unparsable values are ignored.
void apply_tuning_overrides(pt_tuning &t) {
    const char *env = std::getenv("PTGPU_TUNING");
    if (!env) return;
    static const struct { const char *name; uint32_t pt_tuning::*field; } kFields[] = {
        {"pass_form", &pt_tuning::pass_form}, {"pool_refill", &pt_tuning::pool_refill}, {"pool_chunk", &pt_tuning::pool_chunk},
        {"pool_grid", &pt_tuning::pool_grid}, {"tree_refill", &pt_tuning::tree_refill}, {"tree_inner", &pt_tuning::tree_inner},
        {"tree_shade", &pt_tuning::tree_shade}, {"tree_trav_plus1", &pt_tuning::tree_trav_plus1},
        {"accel_form", &pt_tuning::accel_form}, {"accel_refill", &pt_tuning::accel_refill},
        {"accel_inner", &pt_tuning::accel_inner}, {"accel_grid", &pt_tuning::accel_grid}, {"shards", &pt_tuning::shards},
        {"scan_max_ops_plus1", &pt_tuning::scan_max_ops_plus1}, {"list_max_leaves_plus1", &pt_tuning::list_max_leaves_plus1},
        {"profile_period", &pt_tuning::profile_period}};
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string kv = all.substr(pos, end - pos);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos)
            for (const auto &f : kFields)
                if (kv.compare(0, eq, f.name) == 0 && std::strlen(f.name) == eq) {
                    char *stop = nullptr;
                    const unsigned long v = std::strtoul(kv.c_str() + eq + 1, &stop, 10);
                    if (stop && *stop == '\0') t.*(f.field) = (uint32_t)v;
                }
        pos = end + 1;
    }
}

} // namespace

// ================================================================================================================
// extern "C"
// ================================================================================================================
extern "C" {
static int quantise_nodes(pt_ctx *ctx);
static void gather_async_release(pt_ctx *ctx);

int pt_create(const pt_config *cfg, pt_ctx **out) {
    // the structure of ABI version 1 ended before `tuning`: still accepted, with the defaults
    if (!cfg || !out || (cfg->struct_size != sizeof(pt_config) && cfg->struct_size != offsetof(pt_config, tuning)))
        return PT_ERR_INVALID_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return PT_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= n_dev) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = new (std::nothrow) pt_ctx();
    if (!ctx) return PT_ERR_OUT_OF_MEMORY;
    ctx->device = cfg->device;
    ctx->seed = cfg->seed;
    ctx->flags = cfg->flags;
    ctx->exact = (cfg->flags & PT_CFG_EXACT) != 0;
    ctx->compact_spheres = (cfg->flags & PT_CFG_FAT_RECORDS) == 0;
    ctx->quant_pool = (cfg->flags & PT_CFG_WIDE_NODES) == 0;
    pt_tuning tune;
    std::memset(&tune, 0, sizeof(tune));
    if (cfg->struct_size == sizeof(pt_config)) tune = cfg->tuning;
    apply_tuning_overrides(tune); // PTGPU_TUNING, for A/B runs of an unmodified host
    if (!apply_tuning(ctx, tune)) {
        delete ctx;
        return PT_ERR_INVALID_ARG;
    }
    ctx->capacity = cfg->max_paths_in_flight ? cfg->max_paths_in_flight : kDefaultPathsInFlight;
    if (ctx->capacity < 1024) ctx->capacity = 1024;
    auto bail = [&](int code) {
        pt_destroy(ctx);
        return code;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(PT_ERR_HIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(PT_ERR_HIP);
    for (hipEvent_t &e : ctx->ev_batch)
        if (hipEventCreate(&e) != hipSuccess) return bail(PT_ERR_HIP);
    if (static_scratch_for(ctx->device, &ctx->d_static) != hipSuccess) return bail(PT_ERR_HIP);
    int rc = alloc_path_state(ctx);
    if (rc != PT_OK) return bail(rc);
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    gather_async_release(ctx);
    if (ctx->comm && rccl_api()) (void)rccl_api()->CommDestroy(ctx->comm);
    (void)pt_present_detach(ctx);
    void *ptrs[] = {ctx->d_ops,     ctx->d_xforms,  ctx->d_leaves, ctx->d_lights, ctx->d_materials, ctx->d_textures,
                    ctx->d_texels,  ctx->d_perlin,  ctx->d_pool,   ctx->d_queues, ctx->d_counts, ctx->d_stats,     ctx->d_trav,
                    ctx->d_ops_list, ctx->d_ops_fused, ctx->d_accel_nodes, ctx->d_accel_prims, ctx->d_accel_parent, ctx->d_accel_levels,
                    ctx->d_accel_nodes_q, ctx->d_accel_spheres, ctx->d_image_owned, ctx->d_frames, ctx->d_acc[0], ctx->d_acc[1], ctx->d_acc[2], ctx->d_pool_consts,
                    ctx->d_stage[0], ctx->d_stage[1]};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_batch)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

uint64_t pt_device_alloc_count(void) { return g_alloc_count.load(std::memory_order_relaxed); }

const char *pt_last_error(pt_ctx *ctx) { return ctx ? ctx->error.c_str() : "null context"; }

void *pt_stream(pt_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *scene) {
    if (!ctx || !scene) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    std::string err;
    FlatScene flat;
    int rc = flatten_scene(scene, flat, err);
    if (rc != PT_OK) return set_error(ctx, rc, err);
    if (flat.lights.empty()) return set_error(ctx, PT_ERR_INVALID_ARG, "scene has no light-shape list");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = false;
    ctx->flat = std::move(flat);
    const FlatScene &f = ctx->flat;
    // transform / leaf tables are never empty on the device so kernels can index them unconditionally
    std::vector<DevXform> xf = f.xforms;
    if (xf.empty()) xf.push_back(DevXform{0.0, 0.0, 0.0, 0.0, 1.0, 1.0});
    {
        std::vector<DevOp> ops_padded = f.ops; // one sentinel record so that ops[c + 1] is always readable
        DevOp sentinel;
        std::memset(&sentinel, 0, sizeof(sentinel));
        sentinel.kind = op_pack(OP_EXIT, 0, 0);
        ops_padded.push_back(sentinel);
        if ((rc = upload(ctx, &ctx->d_ops, ops_padded.data(), ops_padded.size())) != PT_OK) return rc;
    }
    if ((rc = upload(ctx, &ctx->d_xforms, xf.data(), xf.size())) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_leaves, f.leaves.data(), f.leaves.size())) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_lights, f.lights.data(), f.lights.size())) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_materials, scene->materials, (size_t)scene->n_materials)) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_textures, scene->textures, (size_t)scene->n_textures)) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_texels, scene->texels, (size_t)scene->n_texel_bytes)) != PT_OK) return rc;
    if ((rc = upload(ctx, &ctx->d_perlin, scene->perlin, scene->perlin ? (size_t)1 : (size_t)0)) != PT_OK) return rc;
    SceneView &v = ctx->view;
    v.ops = (const DevOp *)ctx->d_ops;
    v.xforms = (const DevXform *)ctx->d_xforms;
    v.leaves = (const DevLeaf *)ctx->d_leaves;
    v.lights = (const DevLight *)ctx->d_lights;
    v.materials = (const pt_material *)ctx->d_materials;
    v.textures = (const pt_texture *)ctx->d_textures;
    v.texels = (const uint8_t *)ctx->d_texels;
    v.perlin = (const pt_perlin *)ctx->d_perlin;
    v.n_traversal_ops = f.n_traversal_ops;
    v.n_lights = (int32_t)f.lights.size();
    v.n_leaves = (int32_t)f.leaves.size();
    v.has_medium = f.has_medium ? 1 : 0;
    v.has_frames = f.has_frames ? 1 : 0;
    v.zero_can_turn_nan = 0;
    for (int i = 0; i < scene->n_materials; ++i)
        if (scene->materials[i].kind == PT_M_ISOTROPIC) v.zero_can_turn_nan = 1;
    // a Lambertian primitive whose bounds reach inside a sphere light shape: light pdfs from there are NaN
    for (int32_t k = 0; k < f.n_traversal_ops && !v.zero_can_turn_nan; ++k) {
        const DevOp &op = f.ops[k];
        if (op.leaf < 0 || f.leaves[op.leaf].mat_kind != PT_M_LAMBERTIAN) continue;
        double lo[3], hi[3];
        if (!accel_prim_bounds(f, op, lo, hi)) continue;
        for (const DevLight &l : f.lights) {
            if (l.kind != LIGHT_SPHERE) continue;
            double d2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double c = l.d[a], gap = c < lo[a] ? lo[a] - c : (c > hi[a] ? c - hi[a] : 0.0);
                d2 += gap * gap;
            }
            // (the bounds carry the 1e-4 pad of src/rect.rs:116-122: a surface that merely touches the shape, as the
            // Cornell floor touches the glass sphere's, does not count)
            if (std::sqrt(d2) < l.d[3] - 1e-3) v.zero_can_turn_nan = 1;
        }
    }
    // texture level: what the leaves' materials still have to evaluate at a hit (constant colours are in the leaves)
    ctx->tex = 0;
    for (const DevLeaf &lf : f.leaves) {
        if (lf.texture < 0) continue;
        std::vector<int> todo{lf.texture}; // a checker's children count too
        for (size_t k = 0; k < todo.size() && k < 64; ++k) {
            const pt_texture &t = scene->textures[todo[k]];
            if (t.kind == PT_T_NOISE && !scene->perlin)
                return set_error(ctx, PT_ERR_INVALID_ARG, "a NOISE texture needs pt_scene_desc.perlin");
            if (t.kind == PT_T_IMAGE || t.kind == PT_T_NOISE) ctx->tex = 2;
            else if (t.kind != PT_T_CONSTANT) ctx->tex = std::max(ctx->tex, 1);
            if (t.kind == PT_T_CHECKER) {
                todo.push_back(t.even);
                todo.push_back(t.odd);
            }
        }
    }
    // (a medium with a list or tree as boundary is walked by the per-lane kernel alone, whatever the stream's length)
    ctx->use_scan = f.n_traversal_ops <= ctx->scan_max_ops && !f.general_boundaries;
    // In the wave-uniform traversal a primitive is tested as soon as ONE lane of the wave wants it, so a box test
    // only pays when all 64 rays miss the box. For a handful of primitives that almost never happens, and the box
    // tests cost more than the primitive tests they guard. Fast mode therefore scans the primitive ops alone, in the
    // same order; exact mode keeps every box test, i.e. the reference's culling decisions, ray by ray.
    ctx->scan_list = false;
    ctx->fuse_ok = false;
    if (ctx->use_scan && !ctx->exact) {
        std::vector<DevOp> list;
        for (int32_t k = 0; k < f.n_traversal_ops; ++k)
            if (op_kind(f.ops[k].kind) != OP_BOX) list.push_back(f.ops[k]);
        int prims = 0;
        for (const DevOp &op : list) {
            const int kind = op_kind(op.kind);
            if (kind != OP_ENTER && kind != OP_EXIT) ++prims;
        }
        if (prims <= ctx->list_max_leaves && (int)f.leaves.size() <= kLdsLeaves) {
            // MEDIUM ops keep pointing at their boundary ops in the full stream: rebase through an offset table
            for (DevOp &op : list)
                if (op_kind(op.kind) == OP_MEDIUM) op.skip = op.skip - f.n_traversal_ops + (int32_t)list.size() + 1;
            ctx->n_list_ops = (int32_t)list.size();
            DevOp sentinel;
            std::memset(&sentinel, 0, sizeof(sentinel));
            sentinel.kind = op_pack(OP_EXIT, 0, 0);
            list.push_back(sentinel);
            for (size_t k = (size_t)f.n_traversal_ops; k < f.ops.size(); ++k) list.push_back(f.ops[k]); // boundaries
            if ((rc = upload(ctx, &ctx->d_ops_list, list.data(), list.size())) != PT_OK) return rc;
            ctx->scan_list = true;
            // The fused trace+shade kernels read a second form of the list, in which a primitive's transform
            // chain is written out as XFORM ops in front of it: the chain then arrives with the op stream instead
            // of through a dependent load. (MEDIUM ops keep their chain and their boundary-op index.)
            ctx->fuse_ok = !f.has_frames;
            if (ctx->fuse_ok) {
                std::vector<DevOp> fl;
                const size_t n_list = (size_t)ctx->n_list_ops;
                for (size_t k = 0; k < n_list; ++k) {
                    DevOp op = list[k];
                    const int kind = op_kind(op.kind);
                    const int cc = (op.kind >> 16) & 0xff;
                    if (kind != OP_MEDIUM && cc > 0) {
                        for (int j = 0; j < cc; ++j) {
                            const DevXform &x = f.xforms[(size_t)op.chain_start + j];
                            DevOp xo;
                            std::memset(&xo, 0, sizeof(xo));
                            xo.d[0] = x.tx; xo.d[1] = x.ty; xo.d[2] = x.tz;
                            xo.d[3] = x.s; xo.d[4] = x.c; xo.d[5] = x.sign;
                            xo.kind = op_pack(OP_XFORM, 0, 0) | (j == 0 ? kOpFirst : 0);
                            xo.leaf = -1;
                            fl.push_back(xo);
                        }
                        op.kind = (op.kind & 0x0000ffff) | kOpFramed; // the chain is spent
                    }
                    fl.push_back(op);
                }
                ctx->n_fused_ops = (int32_t)fl.size();
                for (DevOp &op : fl) // boundary ops follow the sentinel of THIS list
                    if (op_kind(op.kind) == OP_MEDIUM)
                        op.skip = op.skip - (int32_t)n_list + ctx->n_fused_ops;
                for (size_t k = n_list; k < list.size(); ++k) fl.push_back(list[k]); // sentinel + boundary ops
                if ((rc = upload(ctx, &ctx->d_ops_fused, fl.data(), fl.size())) != PT_OK) return rc;
            }
        }
    }
    // Everything else in fast mode is traced through a SAH tree over the same primitives (accel.h).
    ctx->accel_ok = false;
    if (ctx->d_accel_spheres) {
        (void)hipFree(ctx->d_accel_spheres);
        ctx->d_accel_spheres = nullptr;
    }
    if (!ctx->exact && !ctx->scan_list && ctx->use_accel && !f.general_boundaries) {
        Accel acc;
        std::string why;
        if (build_accel(f, acc, why)) {
            if ((rc = upload(ctx, &ctx->d_accel_nodes, acc.nodes.data(), acc.nodes.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_prims, acc.prims.data(), acc.prims.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_spheres, acc.spheres.data(), acc.spheres.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_parent, acc.parent_slot.data(), acc.parent_slot.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_levels, acc.by_level.data(), acc.by_level.size())) != PT_OK) return rc;
            ctx->accel_levels.clear();
            for (size_t l = 0; l + 1 < acc.level_start.size(); ++l)
                ctx->accel_levels.emplace_back(acc.level_start[l], acc.level_start[l + 1]);
            ctx->accel_depth = acc.depth;
            ctx->n_accel_prims = (int32_t)acc.prims.size();
            ctx->n_accel_nodes = (int32_t)acc.nodes.size();
            if ((rc = quantise_nodes(ctx)) != PT_OK) return rc;
            ctx->accel_ok = true;
            // Camera rays through a small tree take similar numbers of steps, and a ray per thread is then the
            // faster form (measured on ~500 spheres: +9 % on the render); bigger trees gain from the lane pool at
            // every bounce.
            if (!ctx->accel_pool_from_set) ctx->accel_pool_from = acc.nodes.size() <= 1024 ? 1u : 0u;
        }
    }
    ctx->profile.clear();
    ctx->profile_n = 0;
    ctx->have_scene = true;
    return PT_OK;
}

int pt_camera_set(pt_ctx *ctx, const pt_camera *c) {
    if (!ctx || !c) return PT_ERR_INVALID_ARG;
    // Camera::new, src/camera.rs:25-51
    const double theta = c->vfov * kPi / 180.0;
    const double half_height = std::tan(theta / 2.0);
    const double half_width = c->aspect * half_height;
    const D3 origin = hd3(c->origin);
    const D3 w = hunit(hsub(origin, hd3(c->look_at)));
    const D3 u = hunit(hcross(hd3(c->vup), w));
    const D3 v = hcross(w, u);
    CameraDev &cam = ctx->cam;
    cam.origin = origin;
    cam.lower_left_corner = hsub(hsub(hsub(origin, hmul(hmul(u, half_width), c->focus_dist)),
                                      hmul(hmul(v, half_height), c->focus_dist)),
                                 hmul(w, c->focus_dist));
    cam.horizontal = hmul(hmul(hmul(u, 2.0), half_width), c->focus_dist);
    cam.vertical = hmul(hmul(hmul(v, 2.0), half_height), c->focus_dist);
    cam.u = u;
    cam.v = v;
    cam.time0 = c->time0;
    cam.time1 = c->time1;
    cam.lens_radius = c->aperture / 2.0;
    ctx->have_camera = true;
    return PT_OK;
}

int pt_image_bind(pt_ctx *ctx, void *device_rgba, uint32_t width, uint32_t height) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    ctx->d_frames = nullptr;
    if (!device_rgba) {
        ctx->image_external = false;
        ctx->d_image = ctx->d_image_owned;
        if (!ctx->d_image_owned) ctx->width = ctx->height = 0;
        return PT_OK;
    }
    if (width == 0 || height == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "empty image");
    if (!device_range_ok(ctx->device, device_rgba, (size_t)width * height * 4 * sizeof(float)))
        return set_error(ctx, PT_ERR_INVALID_ARG, "device_rgba must be device memory of this context's GPU holding width*height*4 floats");
    if (ctx->d_image_owned) (void)hipFree(ctx->d_image_owned);
    ctx->d_image_owned = nullptr;
    ctx->image_external = true;
    ctx->gathered_slot = -1;
    ctx->d_image = (float *)device_rgba;
    ctx->width = width;
    ctx->height = height;
    return PT_OK;
}

void *pt_image_device_ptr(pt_ctx *ctx) { return ctx ? (void *)ctx->d_image : nullptr; }

int pt_render(pt_ctx *ctx, const pt_render_args *a) {
    if (!ctx || !a || a->struct_size != sizeof(pt_render_args)) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_render before pt_scene_upload");
    if (!ctx->have_camera) return set_error(ctx, PT_ERR_NO_CAMERA, "pt_render before pt_camera_set");
    if (a->width == 0 || a->height == 0 || a->x0 >= a->x1 || a->y0 >= a->y1 || a->x1 > a->width || a->y1 > a->height)
        return set_error(ctx, PT_ERR_INVALID_ARG, "bad tile rectangle");
    if (a->spp == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "spp must be positive");
    if (a->max_depth > (uint32_t)(kMaxDepthSlots - 2)) return set_error(ctx, PT_ERR_TOO_LARGE, "max_depth too large");
    if (a->mode != PT_MODE_OFFLINE && a->mode != PT_MODE_REALTIME) return set_error(ctx, PT_ERR_INVALID_ARG, "bad mode");
    if ((uint64_t)a->width * a->height > 0x7fffffffull) return set_error(ctx, PT_ERR_TOO_LARGE, "image too large");
    (void)hipSetDevice(ctx->device);
    int rc = ensure_image(ctx, a->width, a->height);
    if (rc != PT_OK) return rc;
    ctx->render_flags = a->flags & ~(uint32_t)PT_RENDER_NO_WAIT;

    const uint32_t row_step = a->row_step ? a->row_step : 1u;
    if (a->row_phase >= row_step) return set_error(ctx, PT_ERR_INVALID_ARG, "row_phase must be below row_step");
    const uint32_t span = a->y1 - a->y0;
    if (a->row_phase >= span) return PT_OK; // none of this tile's rows belong to the phase
    // rows y0 + phase + m * step below y1
    const uint32_t tw = a->x1 - a->x0, th = (span - a->row_phase + row_step - 1) / row_step;
    if (tw > ctx->capacity) return set_error(ctx, PT_ERR_TOO_LARGE, "tile row wider than the paths-in-flight budget");
    // chunk = as many whole tile rows as fit the budget with at least one sample each
    const uint32_t rows_per_chunk = std::max<uint32_t>(1, std::min<uint32_t>(th, ctx->capacity / tw));
    rc = ensure_acc(ctx, (size_t)rows_per_chunk * tw);
    if (rc != PT_OK) return rc;
    bool first_of_call = true;
    for (uint32_t r0 = 0; r0 < th; r0 += rows_per_chunk) {
        const uint32_t rows = std::min(rows_per_chunk, th - r0);
        PassParams pp;
        std::memset(&pp, 0, sizeof(pp));
        pp.seed = ctx->seed;
        pp.frame = a->frame;
        pp.width = a->width;
        pp.height = a->height;
        pp.cx0 = a->x0;
        pp.cy0 = a->y0 + a->row_phase + r0 * row_step;
        pp.row_step = row_step;
        pp.cw = tw;
        pp.cpix = rows * tw;
        pp.spp = a->spp;
        pp.max_depth = a->max_depth;
        pp.mode = a->mode;
        pp.flags = a->flags;
        const uint32_t s_per_pass = std::max<uint32_t>(1, std::min<uint32_t>(a->spp, ctx->capacity / pp.cpix));
        for (uint32_t s0 = 0; s0 < a->spp; s0 += s_per_pass) {
            pp.s_base = s0;
            pp.s_count = std::min(s_per_pass, a->spp - s0);
            pp.first_pass = s0 == 0;
            pp.last_pass = s0 + pp.s_count >= a->spp;
            rc = run_pass(ctx, pp, first_of_call);
            first_of_call = false;
            if (rc != PT_OK) return rc;
        }
    }
    if (a->flags & PT_RENDER_NO_WAIT) return PT_OK; // the caller waits when it needs the frame (pt_wait, or any reader)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_wait(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// A batch of tile jobs. The reference cuts the image into tiles because its workers are threads; here a tile is a pass of
// its own (launches, a drain, a synchronisation), and a hundred of them cost a realtime frame thirty times its tracing
// time. Paths are keyed by global pixel and sample, so the pixels do not depend on how the image is cut: a batch whose
// jobs agree in everything but their rectangles, and whose rectangles tile their bounding box exactly, is traced as ONE
// job over that box, bit-identical to the jobs run one by one. Any other batch runs job by job.
int pt_render_jobs(pt_ctx *ctx, const pt_render_args *jobs, size_t n_jobs) {
    if (!ctx || (!jobs && n_jobs)) return PT_ERR_INVALID_ARG;
    if (n_jobs == 0) return PT_OK;
    for (size_t k = 0; k < n_jobs; ++k)
        if (jobs[k].struct_size != sizeof(pt_render_args)) return PT_ERR_INVALID_ARG;
    bool merge = n_jobs > 1;
    pt_render_args box = jobs[0];
    uint64_t area = 0;
    for (size_t k = 0; k < n_jobs && merge; ++k) {
        const pt_render_args &j = jobs[k];
        merge = j.width == box.width && j.height == box.height && j.spp == box.spp && j.max_depth == box.max_depth &&
                j.mode == box.mode && j.frame == box.frame && j.flags == box.flags && j.row_step == box.row_step &&
                j.row_phase == box.row_phase && j.x0 < j.x1 && j.y0 < j.y1;
        if (!merge) break;
        box.x0 = std::min(box.x0, j.x0);
        box.y0 = std::min(box.y0, j.y0);
        box.x1 = std::max(box.x1, j.x1);
        box.y1 = std::max(box.y1, j.y1);
        area += (uint64_t)(j.x1 - j.x0) * (j.y1 - j.y0);
    }
    // disjoint rectangles inside a box whose area they add up to cover it exactly
    merge = merge && area == (uint64_t)(box.x1 - box.x0) * (box.y1 - box.y0) && n_jobs <= 4096;
    for (size_t a = 0; a < n_jobs && merge; ++a)
        for (size_t b = a + 1; b < n_jobs && merge; ++b)
            merge = jobs[a].x1 <= jobs[b].x0 || jobs[b].x1 <= jobs[a].x0 || jobs[a].y1 <= jobs[b].y0 || jobs[b].y1 <= jobs[a].y0;
    // (rows dealt between GPUs start at y0 + row_phase of each job: the jobs' rows are the box's rows only if every
    // job starts on the box's row grid)
    if (merge && box.row_step > 1)
        for (size_t k = 0; k < n_jobs && merge; ++k) merge = (jobs[k].y0 - box.y0) % box.row_step == 0;
    if (merge) return pt_render(ctx, &box);
    for (size_t k = 0; k < n_jobs; ++k)
        if (int rc = pt_render(ctx, &jobs[k]); rc != PT_OK) return rc;
    return PT_OK;
}

int pt_accum_reset(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_image && ctx->width)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_image, 0, (size_t)ctx->width * ctx->height * 4 * sizeof(float), ctx->stream));
    if (ctx->d_frames)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_frames, 0, (size_t)ctx->width * ctx->height * sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_readback(pt_ctx *ctx, float *rgba, size_t n_floats) {
    if (!ctx || !rgba) return PT_ERR_INVALID_ARG;
    if (!ctx->d_image) return set_error(ctx, PT_ERR_INVALID_ARG, "no image has been rendered");
    if (n_floats != (size_t)ctx->width * ctx->height * 4) return set_error(ctx, PT_ERR_INVALID_ARG, "readback size mismatch");
    (void)hipSetDevice(ctx->device);
    const float *image;
    if (int rc = shown_image(ctx, &image); rc != PT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(rgba, image, n_floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = shown_done(ctx); rc != PT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// Extend launches over the rays that sit in slots [0, n) of `q` (one shard: slot = ray index), hits left in
// hit_t / hit_id. `counts` is the queue's fill count (one word on the device), written here. Nothing is allocated.
static int extend_batch(pt_ctx *ctx, const Queue &q, uint32_t shard_cap, uint32_t *counts, double *hit_t, int32_t *hit_id,
                        size_t n, double t_min, double t_max, uint64_t medium_seed, int repeats, double *mean_ms_out) {
    const bool count = (ctx->flags & (PT_CFG_COUNT_TRAVERSAL | PT_CFG_COUNT_ACCEL)) != 0;
    if (counts == ctx->d_counts) ctx->counts_clean_words = 0; // (a frame after this batch clears them itself)
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)counts, (int)(uint32_t)n, 1, ctx->stream));
    const int g = grid_for(ctx, n);
    ExtendArgs ea;
    ea.counts = counts;
    ea.sh = QueueShape{1u, 0u, shard_cap};
    ea.t_min = t_min;
    ea.t_max = t_max;
    ea.medium_seed = medium_seed;
    ea.key_mode = 1;
    ea.depth = 0;
    ea.hit_t = hit_t;
    ea.hit_id = hit_id;
    ea.counters = count ? ctx->d_trav : nullptr;
    if (mean_ms_out) HIP_TRY(ctx, hipEventRecord(ctx->ev_batch[0], ctx->stream));
    for (int r = 0; r < repeats; ++r) launch_extend(ctx, g, q, ea);
    if (mean_ms_out) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_batch[1], ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_batch[0], ctx->ev_batch[1]));
        *mean_ms_out = (double)ms / (double)repeats;
    }
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

static void launch_store_hits(pt_ctx *ctx, const Queue &q, size_t n, const double *hit_t, const int32_t *hit_id, pt_hit *d_hits) {
    const int g = grid_for(ctx, n);
    if (ctx->exact)
        hipLaunchKernelGGL(k_store_hits<true>, dim3(g), dim3(kBlock), 0, ctx->stream, q, (uint32_t)n, ctx->view, hit_t,
                           hit_id, d_hits);
    else
        hipLaunchKernelGGL(k_store_hits<false>, dim3(g), dim3(kBlock), 0, ctx->stream, q, (uint32_t)n, ctx->view, hit_t,
                           hit_id, d_hits);
}

int pt_intersect(pt_ctx *ctx, const pt_ray *rays, size_t n, double t_min, double t_max, uint64_t medium_seed,
                 pt_hit *out) {
    if (!ctx || (n && (!rays || !out))) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_intersect before pt_scene_upload");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    if (n > ctx->capacity) return set_error(ctx, PT_ERR_TOO_LARGE, "ray batch larger than the paths-in-flight budget");
    int rc = ensure_stage(ctx, 0, n * sizeof(pt_ray));
    if (rc == PT_OK) rc = ensure_stage(ctx, 1, n * sizeof(pt_hit));
    if (rc == PT_OK) rc = ensure_queues(ctx);
    if (rc != PT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage[0], rays, n * sizeof(pt_ray), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_load_rays, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, (const pt_ray *)ctx->d_stage[0],
                       (uint32_t)n, ctx->q[0]);
    rc = extend_batch(ctx, ctx->q[0], ctx->capacity, ctx->d_counts, ctx->d_hit_t, ctx->d_hit_id, n, t_min, t_max,
                      medium_seed, 1, nullptr);
    if (rc != PT_OK) return rc;
    launch_store_hits(ctx, ctx->q[0], n, ctx->d_hit_t, ctx->d_hit_id, (pt_hit *)ctx->d_stage[1]);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_stage[1], n * sizeof(pt_hit), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_intersect_static(pt_ctx *ctx, const pt_ray *d_rays, size_t n, double t_min, double t_max, uint64_t medium_seed,
                        pt_hit *d_hits) {
    if (!ctx || (n && (!d_rays || !d_hits))) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_intersect_static before pt_scene_upload");
    if (n == 0) return PT_OK;
    if (n > kStaticMaxRays)
        return set_error(ctx, PT_ERR_TOO_LARGE,
                         "pt_intersect_static takes at most PT_INTERSECT_STATIC_MAX_RAYS (" +
                             std::to_string(kStaticMaxRays) + ") rays per call, got " + std::to_string(n));
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_static) return set_error(ctx, PT_ERR_HIP, "the static scratch buffer was not resolved on this device");
    if (!device_range_ok(ctx->device, d_rays, n * sizeof(pt_ray)) || !device_range_ok(ctx->device, d_hits, n * sizeof(pt_hit)))
        return set_error(ctx, PT_ERR_INVALID_ARG, "d_rays / d_hits must be device memory of this context's GPU holding n entries");
    // the device address resolved by pt_create, never the symbol's name
    unsigned char *const base = ctx->d_static;
    const Queue q{(char *)base};
    double *const hit_t = (double *)(base + kStaticHitTOffset);
    int32_t *const hit_id = (int32_t *)(base + kStaticHitIdOffset);
    uint32_t *const counts = (uint32_t *)(base + kStaticCountsOffset);
    hipLaunchKernelGGL(k_load_rays, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, d_rays, (uint32_t)n, q);
    const int rc = extend_batch(ctx, q, kStaticMaxRays, counts, hit_t, hit_id, n, t_min, t_max, medium_seed, 1, nullptr);
    if (rc != PT_OK) return rc;
    launch_store_hits(ctx, q, n, hit_t, hit_id, d_hits);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_intersect_device(pt_ctx *ctx, const double *d_rays_soa, size_t n, double t_min, double t_max,
                        uint64_t medium_seed, int repeats, double *mean_ms_out) {
    if (!ctx || !d_rays_soa || n == 0 || repeats <= 0) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_intersect_device before pt_scene_upload");
    if (n > ctx->capacity) return set_error(ctx, PT_ERR_TOO_LARGE, "ray batch larger than the paths-in-flight budget");
    (void)hipSetDevice(ctx->device);
    if (!device_range_ok(ctx->device, d_rays_soa, n * 7 * sizeof(double)))
        return set_error(ctx, PT_ERR_INVALID_ARG, "d_rays_soa must be device memory of this context's GPU holding 7 n doubles");
    if (int rc = ensure_queues(ctx); rc != PT_OK) return rc;
    hipLaunchKernelGGL(k_load_rays_soa, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, d_rays_soa, (uint32_t)n,
                       ctx->q[0]);
    double ms = 0.0;
    const int rc = extend_batch(ctx, ctx->q[0], ctx->capacity, ctx->d_counts, ctx->d_hit_t, ctx->d_hit_id, n, t_min, t_max,
                                medium_seed, repeats, &ms);
    if (rc == PT_OK && mean_ms_out) *mean_ms_out = ms;
    return rc;
}

int pt_tonemap_device(pt_ctx *ctx, uint8_t *rgb8_out, size_t n_bytes) {
    if (!ctx || !rgb8_out) return PT_ERR_INVALID_ARG;
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    const size_t need = (size_t)ctx->width * ctx->height * 3;
    if (n_bytes < need) return set_error(ctx, PT_ERR_INVALID_ARG, "rgb8_out too small");
    (void)hipSetDevice(ctx->device);
    const int rc = ensure_stage(ctx, 1, need);
    if (rc != PT_OK) return rc;
    const uint32_t n = ctx->width * ctx->height;
    const float *image;
    if (int rc2 = shown_image(ctx, &image); rc2 != PT_OK) return rc2;
    hipLaunchKernelGGL(k_tonemap_rgb8, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, image, ctx->width,
                       ctx->height, (uint8_t *)ctx->d_stage[1]);
    if (int rc2 = shown_done(ctx); rc2 != PT_OK) return rc2;
    HIP_TRY(ctx, hipMemcpyAsync(rgb8_out, ctx->d_stage[1], need, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_display_transform(pt_ctx *ctx, float exposure, float *rgba_out, size_t n_floats) {
    if (!ctx || !rgba_out) return PT_ERR_INVALID_ARG;
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    const size_t need = (size_t)ctx->width * ctx->height * 4;
    if (n_floats < need) return set_error(ctx, PT_ERR_INVALID_ARG, "rgba_out too small");
    (void)hipSetDevice(ctx->device);
    const int rc = ensure_stage(ctx, 1, need * sizeof(float));
    if (rc != PT_OK) return rc;
    const uint32_t n = ctx->width * ctx->height;
    const float *image;
    if (int rc2 = shown_image(ctx, &image); rc2 != PT_OK) return rc2;
    hipLaunchKernelGGL(k_display_transform<0>, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, image,
                       ctx->width, ctx->height, exposure, ctx->d_stage[1]);
    if (int rc2 = shown_done(ctx); rc2 != PT_OK) return rc2;
    HIP_TRY(ctx, hipMemcpyAsync(rgba_out, ctx->d_stage[1], need * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Presentation ring: frames leave the GPU while the next ones are traced. A slot's frame is first snapshotted on the
// render stream (a device copy, or the display transform written straight into the snapshot), so the image may be
// traced into again at once; the snapshot then crosses PCIe on a second stream into the host's pinned staging memory.
// ---------------------------------------------------------------------------------------------------------------
int pt_present_attach(pt_ctx *ctx, void *host_staging, size_t slot_bytes, uint32_t n_slots) {
    if (!ctx || slot_bytes == 0 || n_slots == 0 || n_slots > pt_ctx::kPresentSlots) return PT_ERR_INVALID_ARG;
    if (ctx->present_host) return set_error(ctx, PT_ERR_INVALID_ARG, "a staging buffer is attached already: pt_present_detach first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!host_staging) {
        // the library's own ring: page-locked host memory the copy engines write at the full rate of the link
        void *ring = nullptr;
        HIP_TRY(ctx, counted_host_malloc(&ring, slot_bytes * n_slots, hipHostMallocDefault));
        host_staging = ring;
        ctx->present_owned = true;
    } else {
        // The caller's memory: whole pages are pinned where the driver allows it. Memory it refuses to pin (some
        // mapped device allocations; every user pointer for a process without the privilege) is still accepted:
        // its copies then pass through the runtime's own bounce buffers, at a fraction of the link's rate.
        const uintptr_t page = 4096, lo = (uintptr_t)host_staging & ~(page - 1);
        const uintptr_t hi = ((uintptr_t)host_staging + slot_bytes * n_slots + page - 1) & ~(page - 1);
        ctx->present_pinned = hipHostRegister((void *)lo, hi - lo, hipHostRegisterDefault) == hipSuccess ? (void *)lo : nullptr;
        (void)hipGetLastError();
    }
    auto fail = [&](hipError_t e, const char *what) {
        ctx->present_host = (unsigned char *)host_staging; // so that detach releases what was made
        (void)pt_present_detach(ctx);
        return set_error(ctx, PT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    if (hipError_t e = hipStreamCreateWithFlags(&ctx->present_stream, hipStreamNonBlocking); e != hipSuccess)
        return fail(e, "hipStreamCreateWithFlags");
    for (uint32_t k = 0; k < n_slots; ++k) {
        if (hipError_t e = hipEventCreateWithFlags(&ctx->ev_snapshot[k], hipEventDisableTiming); e != hipSuccess)
            return fail(e, "hipEventCreateWithFlags");
        if (hipError_t e = hipEventCreateWithFlags(&ctx->ev_landed[k], hipEventDisableTiming); e != hipSuccess)
            return fail(e, "hipEventCreateWithFlags");
    }
    ctx->present_host = (unsigned char *)host_staging;
    ctx->present_slot_bytes = slot_bytes;
    ctx->present_slots = n_slots;
    return PT_OK;
}

void *pt_present_slot(pt_ctx *ctx, uint32_t slot) {
    if (!ctx || !ctx->present_host || slot >= ctx->present_slots) return nullptr;
    return ctx->present_host + (size_t)slot * ctx->present_slot_bytes;
}

int pt_present_pinned(pt_ctx *ctx) { return ctx && ctx->present_host && (ctx->present_owned || ctx->present_pinned) ? 1 : 0; }

int pt_present_wait(pt_ctx *ctx, uint32_t slot) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->present_host || slot >= ctx->present_slots) return set_error(ctx, PT_ERR_INVALID_ARG, "no such presentation slot");
    if (ctx->present_pending[slot]) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_landed[slot]));
        ctx->present_pending[slot] = false;
    }
    return PT_OK;
}

int pt_present_submit(pt_ctx *ctx, uint32_t slot, int kind, float exposure) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->present_host || slot >= ctx->present_slots) return set_error(ctx, PT_ERR_INVALID_ARG, "no such presentation slot");
    if (kind < PT_PRESENT_RAW || kind > PT_PRESENT_DISPLAY8_SRGB) return set_error(ctx, PT_ERR_INVALID_ARG, "unknown pt_present_kind");
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    const bool eight = kind == PT_PRESENT_DISPLAY8 || kind == PT_PRESENT_DISPLAY8_SRGB;
    const size_t need = (size_t)ctx->width * ctx->height * 4 * (eight ? sizeof(uint8_t) : sizeof(float));
    if (need > ctx->present_slot_bytes) return set_error(ctx, PT_ERR_TOO_LARGE, "the image does not fit a presentation slot");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the slot's previous frame must have left its snapshot before the snapshot is overwritten
    if (int rc = pt_present_wait(ctx, slot); rc != PT_OK) return rc;
    if (ctx->present_bytes[slot] < need) {
        if (ctx->d_present[slot]) (void)hipFree(ctx->d_present[slot]);
        ctx->d_present[slot] = nullptr;
        ctx->present_bytes[slot] = 0;
        HIP_TRY(ctx, counted_malloc(&ctx->d_present[slot], need));
        ctx->present_bytes[slot] = need;
    }
    const float *image;
    if (int rc = shown_image(ctx, &image); rc != PT_OK) return rc;
    if (kind == PT_PRESENT_RAW) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_present[slot], image, need, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        const uint32_t n = ctx->width * ctx->height;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, image, ctx->width, ctx->height, exposure,
                               ctx->d_present[slot]);
        };
        if (kind == PT_PRESENT_DISPLAY) launch(k_display_transform<0>);
        else if (kind == PT_PRESENT_DISPLAY8) launch(k_display_transform<1>);
        else launch(k_display_transform<2>);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (int rc = shown_done(ctx); rc != PT_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_snapshot[slot], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->present_stream, ctx->ev_snapshot[slot], 0));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->present_host + (size_t)slot * ctx->present_slot_bytes, ctx->d_present[slot], need,
                                hipMemcpyDeviceToHost, ctx->present_stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_landed[slot], ctx->present_stream));
    ctx->present_pending[slot] = true;
    return PT_OK;
}

int pt_present_detach(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->present_host) return PT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->present_stream) {
        (void)hipStreamSynchronize(ctx->present_stream);
        (void)hipStreamDestroy(ctx->present_stream);
        ctx->present_stream = nullptr;
    }
    if (ctx->present_pinned) (void)hipHostUnregister(ctx->present_pinned);
    if (ctx->present_owned) (void)hipHostFree(ctx->present_host);
    ctx->present_pinned = nullptr;
    ctx->present_owned = false;
    for (uint32_t k = 0; k < pt_ctx::kPresentSlots; ++k) {
        if (ctx->ev_snapshot[k]) (void)hipEventDestroy(ctx->ev_snapshot[k]);
        if (ctx->ev_landed[k]) (void)hipEventDestroy(ctx->ev_landed[k]);
        if (ctx->d_present[k]) (void)hipFree(ctx->d_present[k]);
        ctx->ev_snapshot[k] = ctx->ev_landed[k] = nullptr;
        ctx->d_present[k] = nullptr;
        ctx->present_bytes[k] = 0;
        ctx->present_pending[k] = false;
    }
    ctx->present_host = nullptr;
    ctx->present_slot_bytes = 0;
    ctx->present_slots = 0;
    return PT_OK;
}

// (Re)makes the 32-byte form of the nodes from the f32 nodes on the device; big trees only. The grid spans the root's
// two child boxes, i.e. the scene as the tree sees it now.
static int quantise_nodes(pt_ctx *ctx) {
    if (ctx->d_accel_nodes_q) {
        (void)hipFree(ctx->d_accel_nodes_q);
        ctx->d_accel_nodes_q = nullptr;
    }
    const int32_t n = ctx->n_accel_nodes;
    // big trees, and any tree over compact sphere records (their pass kernel has the registers for the quantised ray)
    if (n < kAccelQuantFrom && ctx->d_accel_spheres == nullptr) return PT_OK;
    AccelNode root;
    HIP_TRY(ctx, hipMemcpyAsync(&root, ctx->d_accel_nodes, sizeof(root), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // A small tree whose root holds a leaf (the scene-wide primitives, csrc/accel.cpp) and the tree over the rest: the grid
    // spans the rest alone, so that a ground sphere of radius 1000 does not set the step for spheres of radius 0.2; rays
    // start on that leaf with the rest pushed, and the root's own boxes, which this grid cannot hold, are never read.
    // (Big trees keep the grid over everything: the extend kernel reads them too, from the root.)
    const bool rest_only = n < kAccelQuantFrom && root.child[0] < 0 && root.child[1] >= 0;
    ctx->accel_q_start[0] = rest_only ? root.child[0] : 0;
    ctx->accel_q_start[1] = rest_only ? root.child[1] : 0;
    AccelGrid g;
    for (int a = 0; a < 3; ++a) {
        const double lo = rest_only ? (double)root.lo[1][a] : std::min((double)root.lo[0][a], (double)root.lo[1][a]);
        const double hi = rest_only ? (double)root.hi[1][a] : std::max((double)root.hi[0][a], (double)root.hi[1][a]);
        if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) return PT_OK; // nothing to quantise against
        g.gmin[a] = lo;
        g.cell[a] = hi > lo ? (hi - lo) / 65535.0 * (1.0 + 1e-12) : 1.0;
    }
    HIP_TRY(ctx, counted_malloc(&ctx->d_accel_nodes_q, (size_t)n * sizeof(AccelNodeQ)));
    hipLaunchKernelGGL(k_accel_quantise, dim3(grid_for(ctx, (uint32_t)n)), dim3(kBlock), 0, ctx->stream,
                       (const AccelNode *)ctx->d_accel_nodes, (AccelNodeQ *)ctx->d_accel_nodes_q, n, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->accel_grid_q = g;
    return PT_OK;
}

// Bounds of every primitive for [time0, time1], pushed up the tree one launch per level, deepest level first.
static int refit_levels(pt_ctx *ctx, double time0, double time1) {
    for (const auto &lv : ctx->accel_levels) {
        const int32_t begin = lv.first, end = lv.second;
        if (end <= begin) continue;
        hipLaunchKernelGGL(k_accel_refit_level, dim3(grid_for(ctx, (uint32_t)(end - begin))), dim3(kBlock), 0, ctx->stream,
                           (AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, ctx->view,
                           (const int32_t *)ctx->d_accel_parent, (const int32_t *)ctx->d_accel_levels, begin, end, time0,
                           time1);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return quantise_nodes(ctx);
}

int pt_scene_refit(pt_ctx *ctx, double time0, double time1) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_scene_refit before pt_scene_upload");
    if (!(time0 <= time1)) return set_error(ctx, PT_ERR_INVALID_ARG, "time0 must not exceed time1");
    if (ctx->exact || (!ctx->accel_ok && !ctx->scan_list))
        return set_error(ctx, PT_ERR_UNSUPPORTED_SCENE,
                         "this context walks the host-built tree of the uploaded scene: rebuild it for the new times");
    if (!ctx->accel_ok) return PT_OK; // primitive list: no boxes to refit
    (void)hipSetDevice(ctx->device);
    return refit_levels(ctx, time0, time1);
}

int pt_scene_build_device(pt_ctx *ctx, double time0, double time1) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_scene_build_device before pt_scene_upload");
    if (!(time0 <= time1)) return set_error(ctx, PT_ERR_INVALID_ARG, "time0 must not exceed time1");
    if (ctx->exact || (!ctx->accel_ok && !ctx->scan_list))
        return set_error(ctx, PT_ERR_UNSUPPORTED_SCENE,
                         "this context walks the host-built tree of the uploaded scene: rebuild it for the new times");
    if (!ctx->accel_ok) return PT_OK; // primitive list: no tree
    (void)hipSetDevice(ctx->device);
    const int32_t n = ctx->n_accel_prims;
    if (n < 2) return set_error(ctx, PT_ERR_UNSUPPORTED_SCENE, "fewer than two primitives");
    if (n >= (1 << 28)) return set_error(ctx, PT_ERR_UNSUPPORTED_SCENE, "a leaf reference holds 28 bits of primitive index");
    // the scene's box for the new window: fit the current tree and read its root
    int rc = refit_levels(ctx, time0, time1);
    if (rc != PT_OK) return rc;
    AccelNode root;
    HIP_TRY(ctx, hipMemcpyAsync(&root, ctx->d_accel_nodes, sizeof(root), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double lo[3], scale[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::min((double)root.lo[0][a], (double)root.lo[1][a]);
        const double extent = std::max((double)root.hi[0][a], (double)root.hi[1][a]) - lo[a];
        scale[a] = extent > 0.0 ? 2097152.0 / extent : 0.0;
    }
    // scratch and the new arrays; whatever is still held here when the function returns is freed
    struct Held {
        void *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *temp = nullptr, *ranges = nullptr, *counter = nullptr;
        void *prims = nullptr, *nodes = nullptr, *parent = nullptr, *spheres = nullptr;
        ~Held() {
            for (void *p : {keys[0], keys[1], vals[0], vals[1], temp, ranges, counter, prims, nodes, parent, spheres})
                if (p) (void)hipFree(p);
        }
    } h;
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(ctx, counted_malloc(&h.keys[k], (size_t)n * sizeof(uint64_t)));
        HIP_TRY(ctx, counted_malloc(&h.vals[k], (size_t)n * sizeof(uint32_t)));
    }
    HIP_TRY(ctx, counted_malloc(&h.prims, (size_t)n * sizeof(DevOp)));
    // at most n - 1 nodes (every leaf a single primitive)
    HIP_TRY(ctx, counted_malloc(&h.nodes, (size_t)n * sizeof(AccelNode)));
    HIP_TRY(ctx, counted_malloc(&h.parent, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, counted_malloc(&h.ranges, (size_t)n * sizeof(BuildRange)));
    HIP_TRY(ctx, counted_malloc(&h.counter, sizeof(int32_t)));
    hipLaunchKernelGGL(k_accel_prim_codes, dim3(grid_for(ctx, (uint32_t)n)), dim3(kBlock), 0, ctx->stream,
                       (const DevOp *)ctx->d_accel_prims, ctx->view, n, time0, time1, D3{lo[0], lo[1], lo[2]},
                       D3{scale[0], scale[1], scale[2]}, (uint64_t *)h.keys[0], (uint32_t *)h.vals[0]);
    size_t temp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint64_t *)h.keys[0], (uint64_t *)h.keys[1],
                                                    (const uint32_t *)h.vals[0], (uint32_t *)h.vals[1], n, 0, 63, ctx->stream));
    HIP_TRY(ctx, counted_malloc(&h.temp, std::max<size_t>(temp_bytes, 16)));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(h.temp, temp_bytes, (const uint64_t *)h.keys[0], (uint64_t *)h.keys[1],
                                                    (const uint32_t *)h.vals[0], (uint32_t *)h.vals[1], n, 0, 63, ctx->stream));
    hipLaunchKernelGGL(k_accel_gather_prims, dim3(grid_for(ctx, (uint64_t)n * 4)), dim3(kBlock), 0, ctx->stream,
                       (const DevOp *)ctx->d_accel_prims, (const uint32_t *)h.vals[1], (DevOp *)h.prims, n);
    if (ctx->d_accel_spheres) { // the compact records follow their primitives (both are 64-byte records)
        static_assert(sizeof(DevSphere) == sizeof(DevOp), "gathered by the same kernel");
        HIP_TRY(ctx, counted_malloc(&h.spheres, (size_t)n * sizeof(DevSphere)));
        hipLaunchKernelGGL(k_accel_gather_prims, dim3(grid_for(ctx, (uint64_t)n * 4)), dim3(kBlock), 0, ctx->stream,
                           (const DevOp *)ctx->d_accel_spheres, (const uint32_t *)h.vals[1], (DevOp *)h.spheres, n);
    }
    // the tree, level by level: node 0 is the root over the whole curve
    const BuildRange whole{0, n};
    const int32_t root_parent = -1, one = 1;
    HIP_TRY(ctx, hipMemcpyAsync(h.ranges, &whole, sizeof(whole), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h.parent, &root_parent, sizeof(root_parent), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h.counter, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
    std::vector<std::pair<int32_t, int32_t>> levels;
    int32_t begin = 0, end = 1;
    for (int32_t level = 0; end > begin; ++level) {
        if (level >= kAccelMaxDepth) return set_error(ctx, PT_ERR_HIP, "device tree build exceeded its depth bound");
        levels.emplace_back(begin, end);
        hipLaunchKernelGGL(k_accel_build_level, dim3(grid_for(ctx, (uint32_t)(end - begin))), dim3(kBlock), 0, ctx->stream,
                           (const uint64_t *)h.keys[1], (AccelNode *)h.nodes, (int32_t *)h.parent, (BuildRange *)h.ranges,
                           begin, end, level, (int32_t *)h.counter);
        int32_t next = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&next, h.counter, sizeof(next), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        begin = end;
        end = next;
        if (end > n) return set_error(ctx, PT_ERR_HIP, "device tree build made more nodes than primitives");
    }
    HIP_TRY(ctx, hipGetLastError());
    // hand the new arrays to the context (the old ones go with `h`) and fit the boxes
    std::swap(ctx->d_accel_prims, h.prims);
    if (ctx->d_accel_spheres) std::swap(ctx->d_accel_spheres, h.spheres);
    std::swap(ctx->d_accel_nodes, h.nodes);
    std::swap(ctx->d_accel_parent, h.parent);
    if (ctx->d_accel_levels) (void)hipFree(ctx->d_accel_levels);
    ctx->d_accel_levels = nullptr; // node ids are level-ordered already
    ctx->accel_levels.assign(levels.rbegin(), levels.rend());
    ctx->accel_depth = (int)levels.size();
    ctx->n_accel_nodes = end; // nodes the build created
    ctx->accel_ok = false; // until the boxes are in
    if ((rc = refit_levels(ctx, time0, time1)) != PT_OK) return rc;
    ctx->accel_ok = true;
    return PT_OK;
}

int pt_last_pass_records(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    return ctx->last_pass_records;
}

int pt_stats_get(pt_ctx *ctx, pt_stats *out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    unsigned long long h[2] = {0, 0};
    TraversalCounters tc{0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&tc, ctx->d_trav, sizeof(tc), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = ctx->stats;
    out->segments = h[0];
    out->nan_pixels = h[1];
    out->node_visits = tc.node_visits;
    out->prim_tests = tc.prim_tests;
    return PT_OK;
}

int pt_stats_reset(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    std::memset(&ctx->stats, 0, sizeof(ctx->stats));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_trav, 0, sizeof(TraversalCounters), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pass_counter = 0;
    return PT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Groups: one context per GPU, each rendering every world-th row of the same image; the rows are gathered to one
// member over RCCL (point-to-point sends: the rows are disjoint, nothing is reduced).
// ---------------------------------------------------------------------------------------------------------------
#define RCCL_OR_FAIL(ctx)                                                                                         \
    const RcclApi *nccl = rccl_api();                                                                             \
    if (!nccl) return set_error(ctx, PT_ERR_COMM, "librccl.so.1 could not be loaded")

#define NCCL_TRY(ctx, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return set_error(ctx, PT_ERR_COMM, std::string(#expr) + ": " + nccl->GetErrorString(r_)); \
    } while (0)

// Inside a GroupStart / GroupEnd pair: a failed call still closes the group before the error is returned, so that the
// communicator is not left collecting operations.
#define NCCL_TRY_IN_GROUP(ctx, expr)                                                                              \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) {                                                                                  \
            (void)nccl->GroupEnd();                                                                               \
            return set_error(ctx, PT_ERR_COMM, std::string(#expr) + ": " + nccl->GetErrorString(r_));             \
        }                                                                                                         \
    } while (0)

static uint32_t rows_of_member(uint32_t height, uint32_t world, uint32_t rank) { return group_rows_of(height, world, rank); }

// The gather for the members of the group that live in this process (all of them, or one).
static int group_gather(pt_ctx *const *m, int n_local, int root) {
    pt_ctx *first = m[0];
    const int world = first->group_world;
    if (root < 0 || root >= world) return set_error(first, PT_ERR_INVALID_ARG, "root outside the group");
    const uint32_t w = first->width, h = first->height;
    for (int k = 0; k < n_local; ++k) {
        pt_ctx *c = m[k];
        if (!c->comm && world > 1) return set_error(c, PT_ERR_INVALID_ARG, "context is not in a group");
        if (!c->d_image || c->width != w || c->height != h || w == 0)
            return set_error(c, PT_ERR_INVALID_ARG, "every member must hold an image of the same size: render first");
        if (c->group_world != world) return set_error(c, PT_ERR_INVALID_ARG, "members of different groups");
    }
    for (int k = 0; k < n_local; ++k) {
        m[k]->gathered_slot = -1; // the rows land in the root's own image
        // transfers of earlier asynchronous gathers on the same communicator come first
        if (m[k]->comm_stream) HIP_TRY(m[k], hipStreamSynchronize(m[k]->comm_stream));
    }
    if (world == 1) return PT_OK;
    RCCL_OR_FAIL(first);
    const size_t row_bytes = (size_t)w * 4 * sizeof(float);
    // stage and pack
    for (int k = 0; k < n_local; ++k) {
        pt_ctx *c = m[k];
        HIP_TRY(c, hipSetDevice(c->device));
        if (c->group_rank == root) {
            size_t rows = 0;
            for (int r = 0; r < world; ++r)
                if (r != root) rows += rows_of_member(h, (uint32_t)world, (uint32_t)r);
            const int rc = ensure_stage(c, 0, std::max<size_t>(rows * row_bytes, 16));
            if (rc != PT_OK) return rc;
        } else {
            const uint32_t rows = rows_of_member(h, (uint32_t)world, (uint32_t)c->group_rank);
            const int rc = ensure_stage(c, 0, std::max<size_t>(rows * row_bytes, 16));
            if (rc != PT_OK) return rc;
            if (rows)
                hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(c, (uint64_t)rows * w)), dim3(kBlock), 0, c->stream,
                                   (const float4 *)c->d_image, w, h, (uint32_t)world, (uint32_t)c->group_rank,
                                   (float4 *)c->d_stage[0]);
        }
    }
    // one group of point-to-point transfers: every peer has its own xGMI link into the root
    NCCL_TRY(first, nccl->GroupStart());
    for (int k = 0; k < n_local; ++k) {
        pt_ctx *c = m[k];
        if (c->group_rank == root) {
            for (int r = 0; r < world; ++r) {
                if (r == root) continue;
                const size_t count = (size_t)rows_of_member(h, (uint32_t)world, (uint32_t)r) * w * 4;
                const size_t offset = (size_t)group_rows_before(h, (uint32_t)world, (uint32_t)root, (uint32_t)r) * w * 4;
                if (count) NCCL_TRY_IN_GROUP(c, nccl->Recv((float *)c->d_stage[0] + offset, count, ncclFloat, r, c->comm, c->stream));
            }
        } else {
            const size_t count = (size_t)rows_of_member(h, (uint32_t)world, (uint32_t)c->group_rank) * w * 4;
            if (count) NCCL_TRY_IN_GROUP(c, nccl->Send(c->d_stage[0], count, ncclFloat, root, c->comm, c->stream));
        }
    }
    NCCL_TRY(first, nccl->GroupEnd());
    for (int k = 0; k < n_local; ++k) {
        pt_ctx *c = m[k];
        HIP_TRY(c, hipSetDevice(c->device));
        if (c->group_rank == root)
            hipLaunchKernelGGL(k_unpack_rows, dim3(grid_for(c, (uint64_t)w * h)), dim3(kBlock), 0, c->stream,
                               (float4 *)c->d_image, w, h, (uint32_t)world, (uint32_t)root, (const float4 *)c->d_stage[0]);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
    }
    return PT_OK;
}

int pt_group_unique_id(uint8_t *id_out, size_t n_bytes) {
    if (!id_out || n_bytes < sizeof(ncclUniqueId)) return PT_ERR_INVALID_ARG;
    const RcclApi *nccl = rccl_api();
    ncclUniqueId id;
    if (!nccl || nccl->GetUniqueId(&id) != ncclSuccess) return PT_ERR_COMM;
    std::memset(id_out, 0, n_bytes);
    std::memcpy(id_out, &id, sizeof(id));
    return PT_OK;
}

int pt_group_join(pt_ctx *ctx, const uint8_t *id, size_t n_bytes, int rank, int world) {
    if (!ctx || !id || n_bytes < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world) return PT_ERR_INVALID_ARG;
    if (ctx->comm) return set_error(ctx, PT_ERR_INVALID_ARG, "context is in a group already: pt_group_leave first");
    RCCL_OR_FAIL(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCL_TRY(ctx, nccl->CommInitRank(&ctx->comm, world, uid, rank));
    ctx->group_rank = rank;
    ctx->group_world = world;
    return PT_OK;
}

int pt_group_join_local(pt_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1) return PT_ERR_INVALID_ARG;
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k]) return PT_ERR_INVALID_ARG;
        if (ctxs[k]->comm) return set_error(ctxs[k], PT_ERR_INVALID_ARG, "context is in a group already");
        for (int j = 0; j < k; ++j)
            if (ctxs[j]->device == ctxs[k]->device)
                return set_error(ctxs[k], PT_ERR_INVALID_ARG, "two members on one GPU: RCCL takes one rank per device");
    }
    RCCL_OR_FAIL(ctxs[0]);
    ncclUniqueId uid;
    NCCL_TRY(ctxs[0], nccl->GetUniqueId(&uid));
    NCCL_TRY(ctxs[0], nccl->GroupStart());
    for (int k = 0; k < n; ++k) {
        if (hipError_t e = hipSetDevice(ctxs[k]->device); e != hipSuccess) {
            (void)nccl->GroupEnd();
            return set_error(ctxs[k], PT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
        }
        NCCL_TRY_IN_GROUP(ctxs[k], nccl->CommInitRank(&ctxs[k]->comm, n, uid, k));
    }
    NCCL_TRY(ctxs[0], nccl->GroupEnd());
    for (int k = 0; k < n; ++k) {
        ctxs[k]->group_rank = k;
        ctxs[k]->group_world = n;
    }
    return PT_OK;
}

int pt_group_layout(uint32_t height, int world, int root, int rank, uint32_t *rows, uint64_t *staged_before) {
    if (world < 1 || root < 0 || root >= world || rank < 0 || rank >= world || !rows || !staged_before) return PT_ERR_INVALID_ARG;
    *rows = group_rows_of(height, (uint32_t)world, (uint32_t)rank);
    *staged_before = group_rows_before(height, (uint32_t)world, (uint32_t)root, (uint32_t)rank);
    return PT_OK;
}

int64_t pt_group_staged_row(uint32_t height, int world, int root, uint32_t y) {
    if (world < 1 || root < 0 || root >= world || y >= height || (int)(y % (uint32_t)world) == root) return -1;
    return (int64_t)group_staged_row(height, (uint32_t)world, (uint32_t)root, y);
}

int pt_group_rows(pt_ctx *ctx, uint32_t *row_step, uint32_t *row_phase) {
    if (!ctx || !row_step || !row_phase) return PT_ERR_INVALID_ARG;
    *row_step = (uint32_t)ctx->group_world;
    *row_phase = (uint32_t)ctx->group_rank;
    return PT_OK;
}

int pt_group_gather(pt_ctx *ctx, int root) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    return group_gather(&ctx, 1, root);
}

int pt_group_gather_local(pt_ctx *const *ctxs, int n, int root) {
    if (!ctxs || n < 1) return PT_ERR_INVALID_ARG;
    for (int k = 0; k < n; ++k)
        if (!ctxs[k]) return PT_ERR_INVALID_ARG;
    if (n != ctxs[0]->group_world) return set_error(ctxs[0], PT_ERR_INVALID_ARG, "pt_group_gather_local takes every member of the group");
    return group_gather(ctxs, n, root);
}

// The asynchronous gather of one member. `loopback`: a virtual group of two on one context (rows 0, 2, ... are the
// root's own, rows 1, 3, ... travel from the context to itself through RCCL), which is what one GPU can check.
static int gather_async(pt_ctx *ctx, int root, bool loopback) {
    const int world = loopback ? 2 : ctx->group_world, rank = loopback ? 0 : ctx->group_rank;
    if (loopback) root = 0;
    if (root < 0 || root >= world) return set_error(ctx, PT_ERR_INVALID_ARG, "root outside the group");
    if (!ctx->comm && (world > 1 || loopback)) return set_error(ctx, PT_ERR_INVALID_ARG, "context is not in a group");
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const RcclApi *nccl = (world > 1) ? rccl_api() : nullptr;
    if (world > 1 && !nccl) return set_error(ctx, PT_ERR_COMM, "librccl.so.1 could not be loaded");
    const uint32_t w = ctx->width, h = ctx->height;
    const size_t row_bytes = (size_t)w * 4 * sizeof(float);
    const bool is_root = rank == root;
    // streams, events and buffers: made once, remade when the image size changes (both streams drained first)
    if (!ctx->comm_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_packed[k], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_gathered[k], hipEventDisableTiming));
        }
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_consumed, hipEventDisableTiming));
    }
    const size_t own_rows = loopback ? h : rows_of_member(h, (uint32_t)world, (uint32_t)rank); // loopback packs both halves
    const size_t need_pack = std::max<size_t>(own_rows * row_bytes, 16);
    const size_t need_recv = is_root ? std::max<size_t>(((size_t)h - rows_of_member(h, (uint32_t)world, (uint32_t)root)) * row_bytes, 16) : 0;
    const size_t need_frame = is_root ? (size_t)h * row_bytes : 0;
    if (need_pack > ctx->gpack_bytes || need_recv > ctx->grecv_bytes || need_frame != ctx->gathered_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
        for (int k = 0; k < 2; ++k) {
            for (void **p : {&ctx->d_gpack[k], &ctx->d_grecv[k], &ctx->d_gathered[k]}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            HIP_TRY(ctx, counted_malloc(&ctx->d_gpack[k], need_pack));
            if (need_recv) HIP_TRY(ctx, counted_malloc(&ctx->d_grecv[k], need_recv));
            if (need_frame) HIP_TRY(ctx, counted_malloc(&ctx->d_gathered[k], need_frame));
            ctx->gather_used[k] = false;
        }
        ctx->gpack_bytes = need_pack;
        ctx->grecv_bytes = need_recv;
        ctx->gathered_bytes = need_frame;
        ctx->gathered_slot = -1;
        ctx->consumed_pending = false;
    }
    const int s = (int)(ctx->gather_seq & 1u);
    hipStream_t R = ctx->stream, C = ctx->comm_stream;
    // render stream: once the slot's previous gather has let go of the staging buffer, pack this member's rows of the frame
    // just traced. From here on the image may be traced into again.
    if (ctx->gather_used[s]) HIP_TRY(ctx, hipStreamWaitEvent(R, ctx->ev_gathered[s], 0));
    float4 *pack = (float4 *)ctx->d_gpack[s];
    const uint32_t own = rows_of_member(h, (uint32_t)world, (uint32_t)rank);
    if (own)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(ctx, (uint64_t)own * w)), dim3(kBlock), 0, R, (const float4 *)ctx->d_image, w, h,
                           (uint32_t)world, (uint32_t)rank, pack);
    float4 *pack_peer = pack + (size_t)own * w; // loopback: the rows of the virtual second member, behind the root's
    const uint32_t peer_rows = loopback ? rows_of_member(h, 2u, 1u) : 0u;
    if (peer_rows)
        hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(ctx, (uint64_t)peer_rows * w)), dim3(kBlock), 0, R, (const float4 *)ctx->d_image, w, h,
                           2u, 1u, pack_peer);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_packed[s], R));
    // comm stream: the transfers, and on the root the assembly of the frame, beside whatever the render stream does next
    HIP_TRY(ctx, hipStreamWaitEvent(C, ctx->ev_packed[s], 0));
    if (is_root && ctx->consumed_pending) HIP_TRY(ctx, hipStreamWaitEvent(C, ctx->ev_consumed, 0)); // a reader of this slot's old frame
    if (world > 1) {
        NCCL_TRY(ctx, nccl->GroupStart());
        if (is_root) {
            for (int r = 0; r < world; ++r) {
                if (r == root) continue;
                const size_t count = (size_t)rows_of_member(h, (uint32_t)world, (uint32_t)r) * w * 4;
                const size_t offset = (size_t)group_rows_before(h, (uint32_t)world, (uint32_t)root, (uint32_t)r) * w * 4;
                if (count) NCCL_TRY_IN_GROUP(ctx, nccl->Recv((float *)ctx->d_grecv[s] + offset, count, ncclFloat, loopback ? 0 : r, ctx->comm, C));
            }
        }
        if (!is_root || loopback) {
            const size_t count = (size_t)(loopback ? peer_rows : own) * w * 4;
            if (count) NCCL_TRY_IN_GROUP(ctx, nccl->Send(loopback ? (const void *)pack_peer : (const void *)pack, count, ncclFloat, loopback ? 0 : root, ctx->comm, C));
        }
        NCCL_TRY(ctx, nccl->GroupEnd());
    }
    if (is_root) {
        float4 *frame = (float4 *)ctx->d_gathered[s];
        if (own)
            hipLaunchKernelGGL(k_unpack_member_rows, dim3(grid_for(ctx, (uint64_t)own * w)), dim3(kBlock), 0, C, frame, w, h, (uint32_t)world,
                               (uint32_t)root, (const float4 *)pack);
        if (world > 1)
            hipLaunchKernelGGL(k_unpack_rows, dim3(grid_for(ctx, (uint64_t)w * h)), dim3(kBlock), 0, C, frame, w, h, (uint32_t)world,
                               (uint32_t)root, (const float4 *)ctx->d_grecv[s]);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_gathered[s], C));
    ctx->gather_used[s] = true;
    ctx->gather_seq += 1;
    if (is_root) ctx->gathered_slot = s;
    return PT_OK;
}

int pt_group_gather_async(pt_ctx *ctx, int root) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    return gather_async(ctx, root, false);
}

int pt_group_loopback_async(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->comm || ctx->group_world != 1) return set_error(ctx, PT_ERR_INVALID_ARG, "needs a group of one");
    return gather_async(ctx, 0, true);
}

int pt_group_wait(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (ctx->gather_seq == 0) return PT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_gathered[(ctx->gather_seq - 1) & 1u]));
    return PT_OK;
}

// Ends the asynchronous gathers of a context: both streams drained, the readers back on the context's own image.
static void gather_async_release(pt_ctx *ctx) {
    if (!ctx->comm_stream) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->comm_stream);
    for (int k = 0; k < 2; ++k) {
        for (void **p : {&ctx->d_gpack[k], &ctx->d_grecv[k], &ctx->d_gathered[k]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        if (ctx->ev_packed[k]) (void)hipEventDestroy(ctx->ev_packed[k]);
        if (ctx->ev_gathered[k]) (void)hipEventDestroy(ctx->ev_gathered[k]);
        ctx->ev_packed[k] = ctx->ev_gathered[k] = nullptr;
        ctx->gather_used[k] = false;
    }
    if (ctx->ev_consumed) (void)hipEventDestroy(ctx->ev_consumed);
    ctx->ev_consumed = nullptr;
    (void)hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
    ctx->gpack_bytes = ctx->grecv_bytes = ctx->gathered_bytes = 0;
    ctx->gather_seq = 0;
    ctx->gathered_slot = -1;
    ctx->consumed_pending = false;
}

int pt_group_leave(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    gather_async_release(ctx);
    if (ctx->comm) {
        RCCL_OR_FAIL(ctx);
        (void)hipSetDevice(ctx->device);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        NCCL_TRY(ctx, nccl->CommDestroy(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->group_rank = 0;
    ctx->group_world = 1;
    return PT_OK;
}

// Loopback check of the gather's data path on ONE GPU: the image's odd rows are packed as member 1 of a group of two
// would pack them, cleared in the image, sent from this context to itself through RCCL, and scattered back as member
// 0 would scatter them. The image must come out unchanged (the caller compares). Needs a group of one (pt_group_join
// with world 1, or pt_group_join_local with one context).
int pt_group_loopback(pt_ctx *ctx) {
    if (!ctx) return PT_ERR_INVALID_ARG;
    if (!ctx->comm || ctx->group_world != 1) return set_error(ctx, PT_ERR_INVALID_ARG, "needs a group of one");
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    RCCL_OR_FAIL(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t w = ctx->width, h = ctx->height;
    const uint32_t rows = rows_of_member(h, 2u, 1u);
    if (rows == 0) return PT_OK;
    const size_t count = (size_t)rows * w * 4;
    int rc = ensure_stage(ctx, 0, count * sizeof(float));
    if (rc == PT_OK) rc = ensure_stage(ctx, 1, count * sizeof(float));
    if (rc != PT_OK) return rc;
    hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(ctx, (uint64_t)rows * w)), dim3(kBlock), 0, ctx->stream,
                       (const float4 *)ctx->d_image, w, h, 2u, 1u, (float4 *)ctx->d_stage[1]);
    for (uint32_t y = 1; y < h; y += 2)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_image + (size_t)y * w * 4, 0xff, (size_t)w * 4 * sizeof(float), ctx->stream));
    NCCL_TRY(ctx, nccl->GroupStart());
    NCCL_TRY_IN_GROUP(ctx, nccl->Send(ctx->d_stage[1], count, ncclFloat, 0, ctx->comm, ctx->stream));
    NCCL_TRY_IN_GROUP(ctx, nccl->Recv(ctx->d_stage[0], count, ncclFloat, 0, ctx->comm, ctx->stream));
    NCCL_TRY(ctx, nccl->GroupEnd());
    hipLaunchKernelGGL(k_unpack_rows, dim3(grid_for(ctx, (uint64_t)w * h)), dim3(kBlock), 0, ctx->stream,
                       (float4 *)ctx->d_image, w, h, 2u, 0u, (const float4 *)ctx->d_stage[0]);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

} // extern "C"
