// Wavefront path tracer for gfx950 (MI355X) and its extern "C" entry points (include/ptgpu.h).
//
// The per-pixel recursion of TraceSceneBatchJob::trace / color (src/trace.rs:117-234) becomes four stages over
// queues of path states held in HBM as structure-of-arrays:
//   raygen   camera samples -> ray queue                      (src/trace.rs:152-158, src/camera.rs:53-59)
//   extend   closest hit of every queued ray                  (Hitable::hit on SceneState.world, src/bvh.rs:74-99)
//   shade    emitted / scatter / light sampling, next ray     (src/trace.rs:205-226, src/material.rs)
//            + compaction of surviving paths into the next queue (wave64 ballot + LDS-staged block offsets)
//   resolve  per-pixel mean in sample order -> f32 RGBA image (src/trace.rs:166-182)
// One pass traces `paths in flight` paths to completion: raygen, then (extend, shade) once per bounce.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "accel.h"
#include "flatten.h"
#include "pt_device.h"

using namespace ptdev;

namespace {

constexpr int kBlock = 256;
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4 // waves per SIMD the fast shade kernels are compiled for (512 / this = VGPR budget)
#endif
constexpr int kWavesPerBlock = kBlock / 64;
constexpr uint32_t kDefaultPathsInFlight = 1u << 22;
constexpr int kMaxDepthSlots = 1024; // bounce slots of counts[]; max_depth is capped at kMaxDepthSlots - 2
constexpr uint32_t kShards = 64;     // queue shards (power of two)

// A queue is cut into `shards` equal regions with a fill count each. Chunk j of 256 paths lives in shard
// j % shards, and a path stays in its shard for life, so compaction is per shard: the one atomic a workgroup
// issues per chunk goes to one of 64 counters instead of a single hot word, and a shard can never overflow
// (paths only die). Consumers walk "virtual chunks" (row c, shard s) up to the fullest shard's row count.
struct QueueShape {
    uint32_t shards;    // power of two, <= 64
    uint32_t shift;     // log2(shards)
    uint32_t shard_cap; // slots per shard, multiple of kBlock
};

// ------------------------------------------------------------------------------------------------------------
// Path-state queues (structure of arrays, one entry per in-flight ray)
// ------------------------------------------------------------------------------------------------------------
// An entry is 96 B: the ray (src/math/ray.rs:3-7), the path throughput, the path's RNG key, the path id within the
// pass and the number of RNG draws consumed so far. Entries are stored in tiles of 64 (one wavefront): inside a tile
// every field is a contiguous run of 64 values, so a wave's access to one field is one fully coalesced 512-B (256-B
// for the two 32-bit fields) transaction, and every field of an entry sits at a compile-time offset from ONE
// per-lane address. That keeps the whole queue behind a single pointer (2 SGPRs instead of 26 for thirteen separate
// arrays) and lets the loads use immediate offsets. A tile is 6 KiB of contiguous memory: exactly six
// 1-KiB direct-to-LDS loads of one wave.
struct Queue {
    char *base;
    static constexpr uint32_t kTileBytes = 64 * 96;
    __device__ __forceinline__ char *at(uint32_t i) const {
        return base + (size_t)(i >> 6) * kTileBytes + (size_t)(i & 63u) * 8u;
    }
    __device__ __forceinline__ char *at32(uint32_t i) const {
        return base + (size_t)(i >> 6) * kTileBytes + 5632u + (size_t)(i & 63u) * 4u;
    }
    __device__ __forceinline__ double &ox(uint32_t i) const { return *(double *)(at(i) + 0 * 512); }
    __device__ __forceinline__ double &oy(uint32_t i) const { return *(double *)(at(i) + 1 * 512); }
    __device__ __forceinline__ double &oz(uint32_t i) const { return *(double *)(at(i) + 2 * 512); }
    __device__ __forceinline__ double &dx(uint32_t i) const { return *(double *)(at(i) + 3 * 512); }
    __device__ __forceinline__ double &dy(uint32_t i) const { return *(double *)(at(i) + 4 * 512); }
    __device__ __forceinline__ double &dz(uint32_t i) const { return *(double *)(at(i) + 5 * 512); }
    __device__ __forceinline__ double &tm(uint32_t i) const { return *(double *)(at(i) + 6 * 512); }
    __device__ __forceinline__ double &bx(uint32_t i) const { return *(double *)(at(i) + 7 * 512); }
    __device__ __forceinline__ double &by(uint32_t i) const { return *(double *)(at(i) + 8 * 512); }
    __device__ __forceinline__ double &bz(uint32_t i) const { return *(double *)(at(i) + 9 * 512); }
    __device__ __forceinline__ uint64_t &key(uint32_t i) const { return *(uint64_t *)(at(i) + 10 * 512); }
    __device__ __forceinline__ uint32_t &pid(uint32_t i) const { return *(uint32_t *)(at32(i)); }
    __device__ __forceinline__ uint32_t &rn(uint32_t i) const { return *(uint32_t *)(at32(i) + 256); }
};

struct CameraDev { // derived fields of Camera::new (src/camera.rs:25-51)
    D3 origin, lower_left_corner, horizontal, vertical, u, v;
    double time0, time1, lens_radius;
};

struct PassParams {
    uint64_t seed;
    uint32_t frame;
    uint32_t width, height; // image size
    uint32_t cx0, cy0;      // first pixel of the chunk
    uint32_t row_step;      // image rows between consecutive chunk rows (1 unless rows are interleaved between GPUs)
    uint32_t cw;            // chunk width in pixels (== tile width)
    uint32_t cpix;          // pixels in the chunk
    uint32_t s_base;        // first sample index of this pass
    uint32_t s_count;       // samples per pixel in this pass
    uint32_t spp;           // samples per pixel of the whole render call
    uint32_t max_depth;
    uint32_t mode;
    uint32_t first_pass, last_pass;
    uint32_t flags;         // pt_render_flags
};

struct TraversalCounters {
    unsigned long long node_visits, prim_tests;
};

// ------------------------------------------------------------------------------------------------------------
// Compaction: one slot per surviving lane. Wave64 ballot + popcount give the rank inside the wave, the four
// wave totals are staged in LDS, and one atomic per workgroup reserves the block's range in the output queue.
// Every thread of the block must call this the same number of times.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t compact_slot(bool alive, uint32_t *out_count, uint32_t *lds /* kWavesPerBlock + 1 */) {
    const unsigned long long ballot = __ballot(alive);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t rank = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            uint32_t c = lds[w];
            lds[w] = total; // exclusive prefix of the wave totals
            total += c;
        }
        lds[kWavesPerBlock] = total ? atomicAdd(out_count, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = lds[kWavesPerBlock] + lds[wave] + rank;
    __syncthreads(); // lds is reused by the next call
    return slot;
}

// A queue's fill counts were written by the launch before this one and do not change while a kernel reads them, so
// they can come through the scalar cache (one s_load for the wave instead of a vector load per lane).
__device__ __forceinline__ uint32_t shard_count(const uint32_t *counts, uint32_t s_id) {
    return *(const uint32_t __attribute__((address_space(4))) *)(counts + s_id);
}

// Largest fill count over the shards of a queue (every lane of every wave gets the value).
__device__ __forceinline__ uint32_t max_shard_count(const uint32_t *counts, uint32_t shards) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v = lane < shards ? counts[lane] : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o > v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ------------------------------------------------------------------------------------------------------------
// raygen
// ------------------------------------------------------------------------------------------------------------
template <bool REALTIME>
__global__ __launch_bounds__(kBlock) void k_raygen(Queue q, QueueShape sh, double *lx, double *ly,
                                                   double *lz, CameraDev cam, PassParams pp, uint32_t *counts) {
    __shared__ uint32_t lds[kWavesPerBlock + 1];
    const uint32_t n = pp.cpix * pp.s_count;
    const uint32_t n_chunks = (n + kBlock - 1) / kBlock;
    for (uint32_t j = blockIdx.x; j < n_chunks; j += gridDim.x) {
        const uint32_t s_id = j & (sh.shards - 1), row = j >> sh.shift;
        const uint32_t pid = j * kBlock + threadIdx.x;
        bool alive = pid < n;
        uint32_t i = 0, jy = 0, s = 0;
        uint64_t pixel = 0;
        if (alive) {
            const uint32_t p = pid % pp.cpix;
            s = pid / pp.cpix;
            i = pp.cx0 + p % pp.cw;
            jy = pp.cy0 + (p / pp.cw) * pp.row_step;
            pixel = (uint64_t)jy * pp.width + i;
            if (REALTIME) alive = !(pixel_skip_draw(pp.seed, pp.frame, pixel) < 0.8); // src/trace.rs:144
        }
        uint32_t local = row * kBlock + threadIdx.x;
        if (REALTIME) local = compact_slot(alive, &counts[s_id], lds);
        const uint32_t slot = s_id * sh.shard_cap + local;
        if (alive) {
            Rng g{path_key(pp.seed, pp.frame, pixel, pp.s_base + s), 0};
            // src/trace.rs:153-156
            double random = g.next();
            const double u = ((double)i + random) / (double)pp.width;
            random = g.next();
            const double v = ((double)jy + random) / (double)pp.height;
            // Camera::get_ray, src/camera.rs:53-59
            D3 rd = random_in_unit_disk(g) * cam.lens_radius;
            D3 offset = cam.u * rd.x + cam.v * rd.y;
            const double time = cam.time0 + g.next() * (cam.time1 - cam.time0);
            D3 direction = cam.lower_left_corner + cam.horizontal * u + cam.vertical * v - cam.origin - offset;
            D3 origin = cam.origin + offset;
            q.ox(slot) = origin.x; q.oy(slot) = origin.y; q.oz(slot) = origin.z;
            q.dx(slot) = direction.x; q.dy(slot) = direction.y; q.dz(slot) = direction.z;
            q.tm(slot) = time;
            q.bx(slot) = 1.0; q.by(slot) = 1.0; q.bz(slot) = 1.0;
            q.pid(slot) = pid;
            q.rn(slot) = g.n;
            q.key(slot) = g.key;
        }
        if (pid < n) { // radiance accumulator of the path, also for skipped pixels
            lx[pid] = 0.0; ly[pid] = 0.0; lz[pid] = 0.0;
        }
    }
    if (!REALTIME && blockIdx.x == 0 && threadIdx.x < sh.shards) {
        // offline: every path is generated, so shard k holds chunks k, k + shards, ... of the pass
        const uint32_t k = threadIdx.x;
        uint32_t fill = 0;
        if (n_chunks > k) {
            const uint32_t mine = (n_chunks - k + sh.shards - 1) >> sh.shift; // chunks of this shard
            fill = mine * kBlock;
            const uint32_t last = k + (mine - 1) * sh.shards;                // its last chunk
            if (last == n_chunks - 1) fill -= n_chunks * kBlock - n;         // the pass's final chunk may be partial
        }
        counts[k] = fill;
    }
}

// ------------------------------------------------------------------------------------------------------------
// extend: stackless traversal of the threaded op stream
// ------------------------------------------------------------------------------------------------------------
struct ExtendArgs {
    const uint32_t *counts; // fill count per shard of the queue
    QueueShape sh;
    double t_min, t_max;
    uint64_t medium_seed;  // key_mode 1: intersect_key(medium_seed, i) (pt_intersect)
    uint32_t key_mode;
    uint32_t depth;
    double *hit_t;
    int32_t *hit_id;
    TraversalCounters *counters; // nullptr unless PT_CFG_COUNT_TRAVERSAL
};

// Per-lane form: every lane walks the stream on its own index. Used when the stream is long (thousands of ops)
// and lanes of a wave visit mostly different nodes.
// FRAMES / MEDIA say whether the stream holds ENTER/EXIT or MEDIUM ops at all: without them the ray stays
// loop-invariant during traversal and the code for those ops is not even compiled in.
template <bool X, bool FRAMES, bool MEDIA>
__global__ __launch_bounds__(kBlock) void k_extend(Queue q, SceneView sv, ExtendArgs a) {
    const bool count = a.counters != nullptr;
    unsigned long long my_nodes = 0, my_prims = 0;
    const uint32_t n_virtual = ((max_shard_count(a.counts, a.sh.shards) + kBlock - 1) / kBlock) << a.sh.shift;
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (a.sh.shards - 1);
        const uint32_t local = (v >> a.sh.shift) * kBlock + threadIdx.x;
        if (local >= a.counts[s_id]) continue;
        const uint32_t i = s_id * a.sh.shard_cap + local;
        const D3 wo = d3(q.ox(i), q.oy(i), q.oz(i));
        const D3 wd = d3(q.dx(i), q.dy(i), q.dz(i));
        const double time = q.tm(i);
        uint64_t pkey = 0;
        if (MEDIA) pkey = (a.key_mode == 0) ? q.key(i) : intersect_key(a.medium_seed, i);

        D3 o = wo, d = wd;
        D3 inv = d3(M<X>::rcp(d.x), M<X>::rcp(d.y), M<X>::rcp(d.z));
        double closest = a.t_max;
        int32_t best = -1;
        int32_t cur = 0;
        const int32_t end = sv.n_traversal_ops;
        while (cur < end) {
            const DevOp *op = &sv.ops[cur];
            const int32_t kword = op->kind;
            const int kind = kword & 0xff;
            if (kind == OP_BOX) {
                if (count) ++my_nodes;
                if (aabb_hit(op->d, o, inv, a.t_min, closest)) cur += 1;
                else cur = op->skip;
                continue;
            }
            if (FRAMES && kind == OP_ENTER) {
                apply_chain<X>(sv.xforms, op->chain_start, (kword >> 16) & 0xff, o, d);
                inv = d3(M<X>::rcp(d.x), M<X>::rcp(d.y), M<X>::rcp(d.z));
            } else if (FRAMES && kind == OP_EXIT) {
                o = wo;
                d = wd;
                apply_chain<X>(sv.xforms, op->chain_start, (kword >> 16) & 0xff, o, d);
                inv = d3(M<X>::rcp(d.x), M<X>::rcp(d.y), M<X>::rcp(d.z));
            } else if (MEDIA && kind == OP_MEDIUM) {
                if (count) my_prims += 2;
                double t;
                if (medium_hit<X>(sv, *op, o, d, inv, time, a.t_min, closest, pkey, a.depth, t)) {
                    closest = t;
                    best = op->leaf;
                }
            } else {
                if (count) my_prims += (kind == OP_BOXPRIM) ? 6 : 1;
                double t;
                int sub;
                if (prim_hit<X>(sv, *op, o, d, inv, time, a.t_min, closest, t, sub)) {
                    closest = t;
                    best = op->leaf | (sub << 28);
                }
            }
            cur += 1;
        }
        a.hit_t[i] = closest;
        a.hit_id[i] = best;
    }
    if (count) {
        atomicAdd(&a.counters->node_visits, my_nodes);
        atomicAdd(&a.counters->prim_tests, my_prims);
    }
}

__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int32_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// Wave-uniform form for short streams (a Cornell box is 15 ops): the wave steps through the stream together.
// Each lane keeps its own "next op" index; the wave executes op c for the lanes whose index is c, where c is the
// smallest index any lane still wants. The op record is therefore fetched once per wave through the scalar cache
// (s_load) into SGPRs instead of once per lane through the vector memory path, all lanes of a wave run the same
// primitive test, and per-lane order (hence every result) is exactly that of the per-lane form.
typedef const DevOp __attribute__((address_space(4))) *ConstOpPtr;

__device__ __forceinline__ DevOp load_op_uniform(const DevOp *p) {
    ConstOpPtr cp = (ConstOpPtr)p;
    DevOp op;
#pragma unroll
    for (int k = 0; k < 6; ++k) op.d[k] = cp->d[k];
    op.kind = cp->kind;
    op.skip = cp->skip;
    op.leaf = cp->leaf;
    op.chain_start = cp->chain_start;
    return op;
}

// Fast-mode slab test: same interval arithmetic as AABB::hit, with the per-axis swap written as min/max.
// (Differs from the reference only when a slab bound is NaN, i.e. the origin lies exactly on a bounding plane
// and the direction is exactly parallel to it.)
__device__ __forceinline__ bool aabb_hit_fast(const double *b, D3 o, D3 inv, double tmin, double tmax) {
    const double x0 = (b[0] - o.x) * inv.x, x1 = (b[3] - o.x) * inv.x;
    const double y0 = (b[1] - o.y) * inv.y, y1 = (b[4] - o.y) * inv.y;
    const double z0 = (b[2] - o.z) * inv.z, z1 = (b[5] - o.z) * inv.z;
    tmin = fmax(tmin, fmin(x0, x1));
    tmax = fmin(tmax, fmax(x0, x1));
    bool ok = tmax > tmin; // the reference returns false as soon as tmax <= tmin on any axis
    tmin = fmax(tmin, fmin(y0, y1));
    tmax = fmin(tmax, fmax(y0, y1));
    ok = ok && (tmax > tmin);
    tmin = fmax(tmin, fmin(z0, z1));
    tmax = fmin(tmax, fmax(z0, z1));
    return ok && (tmax > tmin);
}

// The wave walks the threaded op stream together: op records come through the scalar cache into SGPRs and each lane
// is predicated on "my next op", so results are those of the per-lane walk. (wo, wd) is the world ray; `valid` lanes
// take part. Must be called by all 64 lanes of a wave.
// LIST: the stream holds no BOX ops (see pt_scene_upload), so every lane wants every op and no per-lane index
// is kept at all.
template <bool X, bool FRAMES, bool MEDIA, bool LIST>
__device__ __forceinline__ void extend_scan_walk(const SceneView &sv, D3 wo, D3 wd, double time, uint64_t pkey,
                                                 uint32_t depth, double t_min, double t_max, bool valid, bool count,
                                                 double &closest, int32_t &best, unsigned long long &my_nodes,
                                                 unsigned long long &my_prims) {
    const int32_t end = sv.n_traversal_ops;
    D3 o = wo, d = wd;
    D3 inv = d3(M<X>::rcp(d.x), M<X>::rcp(d.y), M<X>::rcp(d.z));
    closest = t_max;
    best = -1;
    int32_t cur = valid ? 0 : 0x7fffffff;
    int32_t c = -1;
    for (;;) {
        c += 1;
        if (!LIST) {
            if (__ballot(cur == c) == 0ull) c = wave_min_i32(cur); // nobody wants the next op: jump ahead
            c = __builtin_amdgcn_readfirstlane(c);
        }
        if (c >= end) break;
        const DevOp op = load_op_uniform(sv.ops + c); // uniform address: scalar loads into SGPRs
        const int32_t kword = __builtin_amdgcn_readfirstlane(op.kind);
        const int kind = kword & 0xff;
        const bool mine = LIST ? valid : (cur == c);
        if (!LIST && kind == OP_BOX) {
            if (count && mine) ++my_nodes;
            bool hit;
            if constexpr (X) hit = aabb_hit(op.d, o, inv, t_min, closest);
            else hit = aabb_hit_fast(op.d, o, inv, t_min, closest);
            cur = mine ? (hit ? c + 1 : op.skip) : cur;
        } else if (FRAMES && (kind == OP_ENTER || kind == OP_EXIT)) {
            D3 no, nd;
            if (kind == OP_ENTER) {
                no = o;
                nd = d;
            } else { // back to the enclosing frame: world ray, then that frame's chain
                no = wo;
                nd = wd;
            }
            apply_chain<X>(sv.xforms, op.chain_start, (kword >> 16) & 0xff, no, nd);
            if (mine) {
                o = no;
                d = nd;
                inv = d3(M<X>::rcp(d.x), M<X>::rcp(d.y), M<X>::rcp(d.z));
                cur = c + 1;
            }
        } else {
            double t = 0.0;
            int sub = 0;
            bool hit;
            if (MEDIA && kind == OP_MEDIUM) {
                if (count && mine) my_prims += 2;
                hit = medium_hit<X>(sv, op, o, d, inv, time, t_min, closest, pkey, depth, t);
            } else {
                if (count && mine) my_prims += (kind == OP_BOXPRIM) ? 6 : 1;
                hit = prim_hit<X>(sv, op, o, d, inv, time, t_min, closest, t, sub);
            }
            const bool take = mine && hit;
            closest = take ? t : closest;
            best = take ? (op.leaf | (sub << 28)) : best;
            cur = mine ? c + 1 : cur;
        }
    }
}

template <bool X, bool FRAMES, bool MEDIA, bool LIST>
__global__ __launch_bounds__(kBlock) void k_extend_scan(Queue q, SceneView sv, ExtendArgs a) {
    const bool count = a.counters != nullptr;
    unsigned long long my_nodes = 0, my_prims = 0;
    const uint32_t n_virtual = ((max_shard_count(a.counts, a.sh.shards) + kBlock - 1) / kBlock) << a.sh.shift;
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (a.sh.shards - 1);
        const uint32_t ns = a.counts[s_id];
        const uint32_t row0 = (v >> a.sh.shift) * kBlock;
        if (row0 >= ns) continue;
        const uint32_t local = row0 + threadIdx.x;
        const bool valid = local < ns;
        const uint32_t i = s_id * a.sh.shard_cap + local;
        const uint32_t il = valid ? i : s_id * a.sh.shard_cap;
        const D3 o = d3(q.ox(il), q.oy(il), q.oz(il));
        const D3 d = d3(q.dx(il), q.dy(il), q.dz(il));
        const double time = q.tm(il);
        uint64_t pkey = 0;
        if (MEDIA) pkey = (a.key_mode == 0) ? q.key(il) : intersect_key(a.medium_seed, il);
        double closest;
        int32_t best;
        extend_scan_walk<X, FRAMES, MEDIA, LIST>(sv, o, d, time, pkey, a.depth, a.t_min, a.t_max, valid, count, closest, best,
                                                 my_nodes, my_prims);
        if (valid) {
            a.hit_t[i] = closest;
            a.hit_id[i] = best;
        }
    }
    if (count) {
        atomicAdd(&a.counters->node_visits, my_nodes);
        atomicAdd(&a.counters->prim_tests, my_prims);
    }
}

// ---- straight-line (predicated) primitive tests for the fused list walk, fast mode only -----------------------
// The whole wave runs every test, so an early return saves nothing and costs an exec-mask branch; these forms
// compute every lane to the end and return the acceptance as one predicate. Comparisons are written negated
// (!(t < t_min)) where the reference's form lets a NaN through.

// AxisAlignedRect::hit acceptance (src/rect.rs:97-105) for one plane axis
template <int AXIS>
__device__ __forceinline__ bool rect_pred_axis(const double *p, D3 o, D3 d, D3 inv, double t_min, double t_max,
                                               double &t) {
    const double ok = AXIS == 0 ? o.x : (AXIS == 1 ? o.y : o.z);
    const double ik = AXIS == 0 ? inv.x : (AXIS == 1 ? inv.y : inv.z);
    const double oa = AXIS == 0 ? o.y : o.x, da = AXIS == 0 ? d.y : d.x;
    const double ob = AXIS == 2 ? o.y : o.z, db = AXIS == 2 ? d.y : d.z;
    t = (p[4] - ok) * ik;
    const double a = __builtin_fma(t, da, oa), b = __builtin_fma(t, db, ob);
    return !(t < t_min) & !(t > t_max) & !(a < p[0]) & !(a > p[1]) & !(b < p[2]) & !(b > p[3]);
}
__device__ __forceinline__ bool rect_pred(const double *p, int axis, D3 o, D3 d, D3 inv, double t_min, double t_max,
                                          double &t) {
    if (axis == 0) return rect_pred_axis<0>(p, o, d, inv, t_min, t_max, t); // axis is wave-uniform
    if (axis == 1) return rect_pred_axis<1>(p, o, d, inv, t_min, t_max, t);
    return rect_pred_axis<2>(p, o, d, inv, t_min, t_max, t);
}
// Sphere::hit root selection (src/sphere.rs:44-81); dd = d.d and inv_dd = 1/dd are per-ray constants
__device__ __forceinline__ bool sphere_pred(D3 c, double radius, D3 o, D3 d, double dd, double inv_dd, double t_min,
                                            double t_max, double &t, int &which) {
    const D3 oc = o - c;
    const double b = M<false>::dot(oc, d);
    const double cc = __builtin_fma(-radius, radius, M<false>::dot(oc, oc));
    const double disc = __builtin_fma(-dd, cc, b * b);
    const double root = fast_sqrt(disc); // garbage when disc <= 0; the predicate discards it
    const double t0 = (-b - root) * inv_dd, t1 = (-b + root) * inv_dd;
    const bool ok0 = (t0 < t_max) & (t0 > t_min), ok1 = (t1 < t_max) & (t1 > t_min);
    t = ok0 ? t0 : t1;
    which = ok0 ? 0 : 1;
    return (disc > 0.0) & (ok0 | ok1);
}
// AxisAlignedBox::hit (src/axis_aligned_box.rs:17-40): six rectangle tests with a shrinking t_max find the nearer
// of the (at most two) faces the ray crosses inside [t_min, t_max]. The slab form finds the same pair directly:
// entry = latest near plane, exit = earliest far plane; the face is the plane that set it. Ties go to X, then Y,
// then Z, as the reference's face order (+Z -Z +Y -Y +X -X, later face wins an equal t) gives.
__device__ __forceinline__ bool box_pred(const double *pb, D3 o, D3 inv, double t_min, double t_max, double &t,
                                         int &face) {
    const double x0 = (pb[0] - o.x) * inv.x, x1 = (pb[3] - o.x) * inv.x;
    const double y0 = (pb[1] - o.y) * inv.y, y1 = (pb[4] - o.y) * inv.y;
    const double z0 = (pb[2] - o.z) * inv.z, z1 = (pb[5] - o.z) * inv.z;
    const double nx = fmin(x0, x1), fx = fmax(x0, x1);
    const double ny = fmin(y0, y1), fy = fmax(y0, y1);
    const double nz = fmin(z0, z1), fz = fmax(z0, z1);
    const double tn = fmax(nx, fmax(ny, nz)), tf = fmin(fx, fmin(fy, fz));
    const bool entry = !(tn < t_min);
    t = entry ? tn : tf;
    const int axis_n = (nx == tn) ? 0 : ((ny == tn) ? 1 : 2);
    const int axis_f = (fx == tf) ? 0 : ((fy == tf) ? 1 : 2);
    const int axis = entry ? axis_n : axis_f;
    const double ia = axis == 0 ? inv.x : (axis == 1 ? inv.y : inv.z);
    // entering through the max plane means travelling in -axis; leaving through it, in +axis
    const bool max_plane = entry ? (ia < 0.0) : (ia > 0.0);
    face = (2 - axis) * 2 + (max_plane ? 0 : 1);
    // A rectangle accepts t == t_max (src/rect.rs:99), but a box is reached through the BVH node that holds it,
    // whose bounds are exactly the box, and AABB::hit rejects an interval that has shrunk to a point
    // (`tmax <= tmin`, src/math/aabb.rs:41): a box ENTERED exactly at the current closest distance is never
    // tested. That is what decides coplanar pairs such as a box standing on a floor seen from below.
    const bool in_range = entry ? (t < t_max) : !(t > t_max);
    return (tn <= tf) & !(t < t_min) & in_range;
}

// Closest hit over the fused kernels' op list (see pt_scene_upload): no BOX ops, and a primitive's transform chain
// is spelled out as XFORM ops in front of it. Every lane of the wave tests every primitive, in stream order. Op
// records come through the scalar cache, one iteration ahead of their use, so a wave never waits for the record it
// is about to test. Must be called by all 64 lanes of a wave. Fast mode only.
constexpr int kOpFramed = 1 << 24; // fused lists: the primitive is tested against the frame ray set up by XFORM ops
constexpr int kOpFirst = 1 << 25;  // fused lists: this XFORM op starts a chain (from the world ray)

template <bool MEDIA> struct ListWalk {
    const SceneView &sv;
    D3 o, d, inv;    // world ray
    D3 fo, fd, finv; // the ray in the frame of the primitive that follows a run of XFORM ops
    double dd, inv_dd, time, t_min, closest;
    uint64_t pkey;
    uint32_t depth;
    int32_t best;

    __device__ __forceinline__ void test(const DevOp &op, int kword, D3 po, D3 pd, D3 pinv) {
        const int kind = kword & 0xff;
        double t = 0.0;
        int sub = 0;
        bool hit;
        if (kind == OP_RECT) hit = rect_pred(op.d, (kword >> 8) & 3, po, pd, pinv, t_min, closest, t);
        else if (kind == OP_SPHERE)
            hit = sphere_pred(d3(op.d[0], op.d[1], op.d[2]), op.d[3], po, pd, dd, inv_dd, t_min, closest, t, sub);
        else if (kind == OP_BOXPRIM) hit = box_pred(op.d, po, pinv, t_min, closest, t, sub);
        else if (MEDIA && kind == OP_MEDIUM)
            hit = medium_hit<false>(sv, op, o, d, inv, time, t_min, closest, pkey, depth, t);
        else hit = prim_hit_framed<false>(sv, op, po, pd, pinv, time, t_min, closest, t, sub);
        closest = hit ? t : closest;
        best = hit ? (op.leaf | (sub << 28)) : best;
    }
    // kword is wave-uniform, so these are scalar branches between straight-line blocks (no per-lane selects)
    __device__ __forceinline__ void step(const DevOp &op, int kword) {
        if ((kword & 0xff) == OP_XFORM) {
            const DevXform x{op.d[0], op.d[1], op.d[2], op.d[3], op.d[4], op.d[5]};
            if (kword & kOpFirst) {
                fo = o;
                fd = d;
                finv = inv;
            }
            xform_ray<false>(x, fo, fd);
            if (x.s != 0.0) finv = d3(fast_rcp(fd.x), finv.y, fast_rcp(fd.z)); // a Y rotation leaves d.y alone
        } else if (kword & kOpFramed) {
            test(op, kword, fo, fd, finv);
        } else {
            test(op, kword, o, d, inv);
        }
    }
};

// Closest hit over the fused kernels' op list (see pt_scene_upload): no BOX ops, and a primitive's transform chain
// is spelled out as XFORM ops in front of it. Every lane of the wave tests every primitive, in stream order. Op
// records come through the scalar cache, one iteration ahead of their use, so a wave never waits for the record it
// is about to test. Must be called by all 64 lanes of a wave. Fast mode only.
template <bool MEDIA>
__device__ __forceinline__ void extend_list(const SceneView &sv, const DevOp *ops, D3 o, D3 d, double time,
                                            uint64_t pkey, uint32_t depth, double t_min, double t_max, double &closest,
                                            int32_t &best) {
    const D3 inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
    const double dd = M<false>::dot(d, d); // unchanged by the rigid XFORM steps
    ListWalk<MEDIA> w{sv, o, d, inv, o, d, inv, dd, fast_rcp(dd), time, t_min, t_max, pkey, depth, -1};
    // The list ends with a sentinel EXIT op (a fused list holds no other), and the loop runs on the KIND of the
    // record it holds. Scalar loads return out of order, so the compiler can only ever wait for all of them: with
    // the loop test reading the current record, that wait falls before the loop and at the end of each step, after
    // the step's arithmetic has been issued, and never right behind the prefetch of the next record. Two records
    // alternate (a, b) so that no record is ever copied.
    DevOp a = load_op_uniform(ops);
    int ka = __builtin_amdgcn_readfirstlane(a.kind);
    for (int32_t c = 0;; c += 2) {
        if ((ka & 0xff) == OP_EXIT) break;
        const DevOp b = load_op_uniform(ops + c + 1);
        w.step(a, ka);
        const int kb = __builtin_amdgcn_readfirstlane(b.kind);
        if ((kb & 0xff) == OP_EXIT) break;
        a = load_op_uniform(ops + c + 2);
        w.step(b, kb);
        ka = __builtin_amdgcn_readfirstlane(a.kind);
    }
    closest = w.closest;
    best = w.best;
}

// Extend stage on its own over a fused-kernel op list: the same walk the fused trace+shade kernels run, for ray
// batches (pt_intersect) and for passes that keep trace and shade apart.
template <bool MEDIA>
__global__ __launch_bounds__(kBlock) void k_extend_list(Queue q, SceneView sv, ExtendArgs a) {
    const uint32_t n_virtual = ((max_shard_count(a.counts, a.sh.shards) + kBlock - 1) / kBlock) << a.sh.shift;
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (a.sh.shards - 1);
        const uint32_t ns = a.counts[s_id];
        const uint32_t row0 = (v >> a.sh.shift) * kBlock;
        if (row0 >= ns) continue;
        const uint32_t local = row0 + threadIdx.x;
        const bool valid = local < ns;
        const uint32_t i = s_id * a.sh.shard_cap + local;
        const uint32_t il = valid ? i : s_id * a.sh.shard_cap;
        const D3 o = d3(q.ox(il), q.oy(il), q.oz(il));
        const D3 d = d3(q.dx(il), q.dy(il), q.dz(il));
        const double time = q.tm(il);
        uint64_t pkey = 0;
        if (MEDIA) pkey = (a.key_mode == 0) ? q.key(il) : intersect_key(a.medium_seed, il);
        double closest;
        int32_t best;
        extend_list<MEDIA>(sv, sv.ops, o, d, time, pkey, a.depth, a.t_min, a.t_max, closest, best);
        if (valid) {
            a.hit_t[i] = closest;
            a.hit_id[i] = best;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fast-mode traversal of the SAH tree built at upload (accel.h): per-lane, nearer child first, the far child on a
// per-lane stack in LDS. A node record holds the boxes of both children, so one 64-B fetch decides both.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool accel_box(const float *lo, const float *hi, D3 o, D3 inv, double t_min, double t_max,
                                          double &t_near) {
    const double x0 = ((double)lo[0] - o.x) * inv.x, x1 = ((double)hi[0] - o.x) * inv.x;
    const double y0 = ((double)lo[1] - o.y) * inv.y, y1 = ((double)hi[1] - o.y) * inv.y;
    const double z0 = ((double)lo[2] - o.z) * inv.z, z1 = ((double)hi[2] - o.z) * inv.z;
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmax(fmin(z0, z1), t_min));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmin(fmax(z0, z1), t_max));
    t_near = tn;
    return tn <= tf;
}

template <bool MEDIA>
__global__ __launch_bounds__(kBlock, MEDIA ? 4 : 5) void k_extend_accel(Queue q, SceneView sv, const AccelNode *nodes,
                                                         const DevOp *prims, ExtendArgs a) {
    __shared__ int32_t stack[kAccelMaxDepth][kBlock]; // [depth][thread]: conflict-free
    const uint32_t n_virtual = ((max_shard_count(a.counts, a.sh.shards) + kBlock - 1) / kBlock) << a.sh.shift;
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (a.sh.shards - 1);
        const uint32_t local = (v >> a.sh.shift) * kBlock + threadIdx.x;
        if (local >= shard_count(a.counts, s_id)) continue;
        const uint32_t i = s_id * a.sh.shard_cap + local;
        const D3 o = d3(q.ox(i), q.oy(i), q.oz(i));
        const D3 d = d3(q.dx(i), q.dy(i), q.dz(i));
        const double time = q.tm(i);
        uint64_t pkey = 0;
        if (MEDIA) pkey = (a.key_mode == 0) ? q.key(i) : intersect_key(a.medium_seed, i);
        const D3 inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
        double closest = a.t_max;
        int32_t best = -1;
        // tests the primitives of a leaf child
        auto leaf = [&](int32_t first, int32_t count) {
            for (int32_t k = first; k < first + count; ++k) {
                const DevOp op = prims[k];
                double t = 0.0;
                int sub = 0;
                bool hit;
                if (MEDIA && (op.kind & 0xff) == OP_MEDIUM)
                    hit = medium_hit<false>(sv, op, o, d, inv, time, a.t_min, closest, pkey, a.depth, t);
                else hit = prim_hit<false>(sv, op, o, d, inv, time, a.t_min, closest, t, sub);
                if (hit) {
                    closest = t;
                    best = op.leaf | (sub << 28);
                }
            }
        };
        int32_t node = 0;
        int sp = 0;
        // A ray with a NaN component passes EVERY slab test (min / max drop NaNs; the reference's comparisons all
        // come out false, src/math/aabb.rs:27-47), i.e. it would walk the whole tree, a million nodes, to find at
        // best a NaN hit distance. Such rays exist by the reference's own semantics (a light sample taken from
        // inside the sphere light shape, src/sphere.rs:92-100) and their paths are NaN whatever they hit, so fast
        // mode does not trace them: they miss.
        const double probe = (o.x + o.y + o.z) + (d.x + d.y + d.z);
        const bool traceable = probe == probe && fabs(probe) <= kF64Max;
        while (traceable) {
            const AccelNode &n = nodes[node];
            double t0, t1;
            bool h0 = accel_box(n.lo[0], n.hi[0], o, inv, a.t_min, closest, t0);
            bool h1 = accel_box(n.lo[1], n.hi[1], o, inv, a.t_min, closest, t1);
            const int32_t c0 = n.child[0], c1 = n.child[1];
            // leaf children are tested on the spot, nearer one first so that the farther may be skipped
            if (h0 && h1 && c0 < 0 && c1 < 0 && t1 < t0) {
                leaf((~c1) >> 3, (~c1) & 7);
                h1 = false;
            }
            if (h0 && c0 < 0) {
                if (t0 <= closest) leaf((~c0) >> 3, (~c0) & 7);
                h0 = false;
            }
            if (h1 && c1 < 0) {
                if (t1 <= closest) leaf((~c1) >> 3, (~c1) & 7);
                h1 = false;
            }
            h0 = h0 && t0 <= closest;
            h1 = h1 && t1 <= closest;
            if (h0 && h1) {
                const bool first0 = t0 <= t1;
                stack[sp++][threadIdx.x] = first0 ? c1 : c0;
                node = first0 ? c0 : c1;
            } else if (h0) {
                node = c0;
            } else if (h1) {
                node = c1;
            } else {
                if (sp == 0) break;
                node = stack[--sp][threadIdx.x];
            }
        }
        a.hit_t[i] = closest;
        a.hit_id[i] = best;
    }
}

// The same traversal with the wave kept full. Rays of one wave need very different numbers of steps (a ray that
// leaves the scene at once next to one that grazes a thousand boxes), so in the form above most lanes of a wave sit
// idle for most of its life. Here a wave owns a run of the queue and works as a pool of 64 ray slots: a lane whose
// ray is done writes its result and, once enough lanes are free, the free lanes take the next rays of the run
// (ballot + prefix count, no atomics: the run belongs to this wave alone). Each round first steps every lane down
// through inner nodes until it stands on a leaf (or is done), then tests the leaves, so both loops run with most
// lanes active. Runs are dealt shard by shard; any grid is correct.
template <bool MEDIA>
__global__ __launch_bounds__(kBlock, MEDIA ? 4 : 5) void k_extend_accel_pool(Queue q, SceneView sv, const AccelNode *nodes,
                                                                            const DevOp *prims, ExtendArgs a,
                                                                            uint32_t tuning) {
    __shared__ int32_t stack[kAccelMaxDepth][kBlock]; // [depth][thread]: conflict-free
    const uint32_t refill_at = tuning & 0xffu, inner_min = tuning >> 8;
    constexpr int32_t kDone = INT32_MIN;             // never a leaf reference (those are ~((first << 3) | count))
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    // virtual waves: a multiple of the shard count, at least one per shard; wave w serves w, w + n_waves, ...
    const uint32_t per_shard = (n_waves + a.sh.shards - 1) >> a.sh.shift;
    const uint32_t n_virtual = per_shard << a.sh.shift;
    uint32_t v_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    uint32_t cur = 0, end = 0, base = 0; // the run being served: slots [cur, end) of the shard starting at `base`
    bool has = false;                    // this lane holds a ray in flight
    uint32_t my_i = 0;
    D3 o = d3(0, 0, 0), d = o, inv = o;
    double time = 0.0, closest = 0.0;
    uint64_t pkey = 0;
    int32_t best = -1, node = kDone;
    int sp = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        if (idle) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (cur >= end) {
                if (v_next < n_virtual) { // next run of this wave
                    const uint32_t s_id = v_next & (a.sh.shards - 1), k = v_next >> a.sh.shift;
                    const uint32_t count = shard_count(a.counts, s_id);
                    const uint32_t per = (((count + per_shard - 1) / per_shard) + 63u) & ~63u;
                    cur = min(k * per, count);
                    end = min(cur + per, count);
                    base = s_id * a.sh.shard_cap;
                    v_next += n_waves;
                    continue;
                }
                if (n_idle == 64) break; // nothing in flight and nothing left to take
            } else if (n_idle >= refill_at || n_idle == 64) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!has && cur + rank < end) {
                    my_i = base + cur + rank;
                    o = d3(q.ox(my_i), q.oy(my_i), q.oz(my_i));
                    d = d3(q.dx(my_i), q.dy(my_i), q.dz(my_i));
                    time = q.tm(my_i);
                    if (MEDIA) pkey = (a.key_mode == 0) ? q.key(my_i) : intersect_key(a.medium_seed, my_i);
                    inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
                    closest = a.t_max;
                    best = -1;
                    sp = 0;
                    // rays with a NaN or infinite component are not traced (see k_extend_accel)
                    const double probe = (o.x + o.y + o.z) + (d.x + d.y + d.z);
                    node = (probe == probe && fabs(probe) <= kF64Max) ? 0 : kDone;
                    has = true;
                }
                cur += n_idle;
            }
        }
        // down through inner nodes, nearer child first, the farther one on the stack, for as long as at least
        // `inner_min` lanes are still on inner nodes
        for (;;) {
            if (has && node >= 0) {
                const AccelNode &n = nodes[node];
                double t0, t1;
                const bool h0 = accel_box(n.lo[0], n.hi[0], o, inv, a.t_min, closest, t0);
                const bool h1 = accel_box(n.lo[1], n.hi[1], o, inv, a.t_min, closest, t1);
                const int32_t c0 = n.child[0], c1 = n.child[1];
                if (h0 && h1) {
                    const bool first0 = t0 <= t1;
                    stack[sp++][threadIdx.x] = first0 ? c1 : c0;
                    node = first0 ? c0 : c1;
                } else if (h0 || h1) {
                    node = h0 ? c0 : c1;
                } else {
                    node = sp ? stack[--sp][threadIdx.x] : kDone;
                }
            }
            const uint32_t n_inner = (uint32_t)__popcll(__ballot(has && node >= 0));
            if (n_inner == 0 || n_inner < inner_min) break;
        }
        // the primitives of a leaf
        if (has && node < 0 && node != kDone) {
            const int32_t ref = ~node;
            const int32_t first = ref >> 3, count = ref & 7;
            for (int32_t k = first; k < first + count; ++k) {
                const DevOp op = prims[k];
                double t = 0.0;
                int sub = 0;
                bool hit;
                if (MEDIA && (op.kind & 0xff) == OP_MEDIUM)
                    hit = medium_hit<false>(sv, op, o, d, inv, time, a.t_min, closest, pkey, a.depth, t);
                else hit = prim_hit<false>(sv, op, o, d, inv, time, a.t_min, closest, t, sub);
                if (hit) {
                    closest = t;
                    best = op.leaf | (sub << 28);
                }
            }
            node = sp ? stack[--sp][threadIdx.x] : kDone;
        }
        if (has && node == kDone) {
            a.hit_t[my_i] = closest;
            a.hit_id[my_i] = best;
            has = false;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Refit of the fast-mode tree for a new shutter window: the bounds of every primitive are recomputed on the device
// (the same rules as accel.cpp's local_bounds / unxform_box) and pushed up level by level, deepest level first.
// ------------------------------------------------------------------------------------------------------------
struct BoxD {
    double lo[3], hi[3];
};
__device__ __forceinline__ void box_clear(BoxD &b) {
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = kF64Max;
        b.hi[k] = -kF64Max;
    }
}
__device__ __forceinline__ void box_grow(BoxD &b, const BoxD &o) {
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = fmin(b.lo[k], o.lo[k]);
        b.hi[k] = fmax(b.hi[k], o.hi[k]);
    }
}
// p = R^T p' + t for the eight corners (the step maps p' = R (p - t), see xform_ray)
__device__ __forceinline__ BoxD box_unxform(const DevXform &x, const BoxD &b) {
    BoxD out;
    box_clear(out);
    for (int corner = 0; corner < 8; ++corner) {
        const double px = (corner & 1) ? b.hi[0] : b.lo[0];
        const double py = (corner & 2) ? b.hi[1] : b.lo[1];
        const double pz = (corner & 4) ? b.hi[2] : b.lo[2];
        const double p[3] = {x.c * px + x.s * pz + x.tx, py + x.ty, -x.s * px + x.c * pz + x.tz};
        for (int k = 0; k < 3; ++k) {
            out.lo[k] = fmin(out.lo[k], p[k]);
            out.hi[k] = fmax(out.hi[k], p[k]);
        }
    }
    return out;
}
__device__ bool prim_local_bounds(const SceneView &sv, const DevOp &op, double t0, double t1, BoxD &b) {
    const int kind = op.kind & 0xff;
    const double *d = op.d;
    if (kind == OP_SPHERE) {
        const double r = fabs(d[3]);
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = d[k] - r;
            b.hi[k] = d[k] + r;
        }
        return true;
    }
    if (kind == OP_MSPHERE) { // union of the boxes at both ends of the window (src/sphere.rs:177-183)
        const double *e = sv.leaves[op.leaf].e;
        const double r = fabs(d[3]);
        box_clear(b);
        for (int w = 0; w < 2; ++w) {
            const double f = ((w ? t1 : t0) - d[4]) / d[5];
            for (int k = 0; k < 3; ++k) {
                const double c = d[k] + f * e[k];
                b.lo[k] = fmin(b.lo[k], c - r);
                b.hi[k] = fmax(b.hi[k], c + r);
            }
        }
        return true;
    }
    if (kind == OP_RECT) {
        const int axis = (op.kind >> 8) & 3;
        const int a_axis = axis == 0 ? 1 : 0, b_axis = axis == 2 ? 1 : 2;
        b.lo[a_axis] = d[0];
        b.hi[a_axis] = d[1];
        b.lo[b_axis] = d[2];
        b.hi[b_axis] = d[3];
        b.lo[axis] = d[4] - 0.0001;
        b.hi[axis] = d[4] + 0.0001;
        return true;
    }
    if (kind == OP_BOXPRIM) {
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = d[k] - 0.0001;
            b.hi[k] = d[k + 3] + 0.0001;
        }
        return true;
    }
    return false;
}
__device__ BoxD prim_world_bounds(const SceneView &sv, const DevOp &op, double t0, double t1) {
    BoxD b;
    box_clear(b);
    if ((op.kind & 0xff) == OP_MEDIUM) { // the boundary primitive through the boundary's own chain
        const DevOp bop = sv.ops[op.skip];
        if (prim_local_bounds(sv, bop, t0, t1, b)) {
            const int cc = (bop.kind >> 16) & 0xff;
            for (int k = cc - 1; k >= 0; --k) b = box_unxform(sv.xforms[bop.chain_start + k], b);
        }
    } else {
        (void)prim_local_bounds(sv, op, t0, t1, b);
    }
    const int cc = (op.kind >> 16) & 0xff; // a tree primitive carries its full world -> primitive chain
    for (int k = cc - 1; k >= 0; --k) b = box_unxform(sv.xforms[op.chain_start + k], b);
    return b;
}
__global__ __launch_bounds__(kBlock) void k_accel_refit_level(AccelNode *nodes, const DevOp *prims, SceneView sv,
                                                              const int32_t *parent_slot, const int32_t *by_level,
                                                              int32_t begin, int32_t end, double t0, double t1) {
    for (int32_t idx = begin + (int32_t)(blockIdx.x * kBlock + threadIdx.x); idx < end; idx += (int32_t)(gridDim.x * kBlock)) {
        const int32_t id = by_level ? by_level[idx] : idx; // a device-built tree numbers its nodes level by level
        AccelNode &n = nodes[id];
        BoxD whole;
        box_clear(whole);
        for (int s = 0; s < 2; ++s) {
            const int32_t c = n.child[s];
            BoxD b;
            if (c < 0) { // leaf child: recompute from its primitives, store rounded outward
                box_clear(b);
                const int32_t first = (~c) >> 3, count = (~c) & 7;
                for (int32_t k = first; k < first + count; ++k) box_grow(b, prim_world_bounds(sv, prims[k], t0, t1));
                for (int a = 0; a < 3; ++a) {
                    n.lo[s][a] = __double2float_rd(b.lo[a]);
                    n.hi[s][a] = __double2float_ru(b.hi[a]);
                }
            }
            for (int a = 0; a < 3; ++a) { // inner child: its level ran before this one and wrote these
                b.lo[a] = (double)n.lo[s][a];
                b.hi[a] = (double)n.hi[s][a];
            }
            box_grow(whole, b);
        }
        const int32_t ps = parent_slot[id];
        if (ps >= 0) {
            AccelNode &p = nodes[ps >> 1];
            const int s = ps & 1;
            for (int a = 0; a < 3; ++a) {
                p.lo[s][a] = (float)whole.lo[a]; // exact: unions of f32 values
                p.hi[s][a] = (float)whole.hi[a];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Tree build on the device (pt_scene_build_device): Morton codes of the primitives' centres, a radix sort, a gather
// of the 64-B primitive records into curve order; the boxes then come from the refit kernels above.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t morton_spread21(uint64_t v) { // bit k of a 21-bit value moves to bit 3 k
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__global__ __launch_bounds__(kBlock) void k_accel_prim_codes(const DevOp *prims, SceneView sv, int32_t n, double t0,
                                                             double t1, D3 lo, D3 scale, uint64_t *keys, uint32_t *vals) {
    for (int32_t k = (int32_t)(blockIdx.x * kBlock + threadIdx.x); k < n; k += (int32_t)(gridDim.x * kBlock)) {
        const BoxD b = prim_world_bounds(sv, prims[k], t0, t1);
        uint64_t code = 0;
        const double s[3] = {scale.x, scale.y, scale.z}, l[3] = {lo.x, lo.y, lo.z};
        for (int a = 0; a < 3; ++a) {
            const double q = (0.5 * (b.lo[a] + b.hi[a]) - l[a]) * s[a]; // [0, 2^21) across the scene's box
            const uint64_t cell = (uint64_t)fmin(fmax(q, 0.0), 2097151.0);
            code |= morton_spread21(cell) << a;
        }
        keys[k] = code;
        vals[k] = (uint32_t)k;
    }
}
// out[j] = in[order[j]], 16 bytes per thread (a record is four of them)
__global__ __launch_bounds__(kBlock) void k_accel_gather_prims(const DevOp *in, const uint32_t *order, DevOp *out, int32_t n) {
    const int64_t total = (int64_t)n * 4;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int32_t j = (int32_t)(t >> 2), part = (int32_t)(t & 3);
        ((uint4 *)(out + j))[part] = ((const uint4 *)(in + order[j]))[part];
    }
}

struct BuildRange {
    int32_t first, n; // the primitives, in curve order, below a node
};
// One level of the tree: every node of the level splits its run of the sorted codes and creates its children. A child
// of at most four primitives is a leaf reference; a larger one becomes a node of the next level, numbered by an
// atomic counter (so a level's nodes are consecutive). The split is where the highest differing code bit flips, as
// long as a balanced subtree below it would still fit the traversal stack; after that, and between equal codes, it
// is the middle of the run.
__global__ __launch_bounds__(kBlock) void k_accel_build_level(const uint64_t *keys, AccelNode *nodes, int32_t *parent_slot,
                                                              BuildRange *ranges, int32_t begin, int32_t end,
                                                              int32_t level, int32_t *next_id) {
    for (int32_t j = begin + (int32_t)(blockIdx.x * kBlock + threadIdx.x); j < end; j += (int32_t)(gridDim.x * kBlock)) {
        const int32_t first = ranges[j].first, n = ranges[j].n, last = first + n - 1;
        int log2n = 0;
        while ((1 << log2n) < n) ++log2n;
        const uint64_t a = keys[first], b = keys[last];
        int32_t split = first + n / 2;
        if (a != b && level + log2n + 2 < kAccelMaxDepth - 2) {
            const int bit = 63 - __clzll((long long)(a ^ b));
            int32_t lo = first, hi = last; // bit clear at lo, set at hi
            while (hi - lo > 1) {
                const int32_t mid = lo + (hi - lo) / 2;
                if ((keys[mid] >> bit) & 1ull) hi = mid;
                else lo = mid;
            }
            split = hi;
        }
        AccelNode &node = nodes[j];
        node.pad[0] = node.pad[1] = 0;
        for (int s = 0; s < 2; ++s) {
            const int32_t cf = s ? split : first, cn = s ? first + n - split : split - first;
            int32_t ref;
            if (cn <= 4) {
                ref = ~((cf << 3) | cn);
            } else {
                ref = atomicAdd(next_id, 1);
                ranges[ref] = BuildRange{cf, cn};
                parent_slot[ref] = j * 2 + s;
            }
            node.child[s] = ref;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// shade + compaction
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void add_radiance(double *lx, double *ly, double *lz, uint32_t pid, D3 c) {
    // Adding an exact zero changes nothing; a NaN compares unequal to zero and is added, which keeps the
    // reference's NaN propagation (0 * x / 0 in src/trace.rs:217-220).
    // A path id belongs to one lane of one launch at a time, so the hardware f64 add in L2 gives the same sum as a
    // load / add / store here, without making the wave wait for the load.
    if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) {
        unsafeAtomicAdd(&lx[pid], c.x);
        unsafeAtomicAdd(&ly[pid], c.y);
        unsafeAtomicAdd(&lz[pid], c.z);
    }
}

// One path vertex: the body of `color` (src/trace.rs:198-234) for a ray whose closest hit is (ht, id), id < 0 for a
// miss. `lf` is the hit primitive's leaf record (fields are fetched where they are used; any record when id < 0).
// Adds the vertex's radiance to the path's accumulator and, if the path goes on, returns true with the next ray,
// throughput and RNG draw count.
// FIRST: the vertex is the path's first one and nothing has initialised the path's accumulator: the vertex's
// radiance (every branch adds at most once) is stored, zero included, instead of added.
// RETURN: the vertex's radiance is handed back in `vertex_radiance` and nothing is written (the caller keeps the
// path's sum).
enum RadianceMode { RAD_ADD = 0, RAD_FIRST = 1, RAD_RETURN = 2 };
template <bool X, bool NEED_UV, int FIRST = RAD_ADD>
__device__ __forceinline__ bool shade_vertex(const SceneView &sv, const DevLeaf &lf, D3 o, D3 d, double time, D3 beta,
                                             uint64_t pkey, uint32_t pid, uint32_t &rn, double ht, int32_t id,
                                             uint32_t depth, uint32_t max_depth, uint32_t flags, double *lx, double *ly,
                                             double *lz, D3 &no, D3 &nd, D3 &nbeta, D3 *vertex_radiance = nullptr) {
    bool alive = false;
    D3 first = d3(0.0, 0.0, 0.0);
    auto add_radiance = [&](double *ax, double *ay, double *az, uint32_t path, D3 c) {
        if constexpr (FIRST != RAD_ADD) first = c;
        else ::add_radiance(ax, ay, az, path, c);
    };
    if (id < 0) {
        // miss: colour is black (src/trace.rs:227); the product still carries a NaN throughput through
        add_radiance(lx, ly, lz, pid, beta * 0.0);
    } else {
        Rng g{pkey, rn};
        const HitRec rec = finalize_hit<X, NEED_UV>(sv, lf, o, d, time, ht, (id >> 28) & 7);
        const D3 emissive = emitted(sv, lf, d, rec);
        bool done = true, dead = false;
        if (depth < max_depth) {
            if (lf.mat_kind == PT_M_DIELECTRIC) { // src/material.rs:177-212
                D3 outward_normal;
                D3 reflected = reflect<X>(d, rec.n);
                double ni_over_nt, reflect_prob, cosine;
                D3 refracted = d3(0.0, 0.0, 0.0);
                const double d_dot_n = M<X>::dot(d, rec.n);
                if (d_dot_n > 0.0) {
                    outward_normal = -rec.n;
                    ni_over_nt = lf.ref_idx;
                    cosine = M<X>::div(lf.ref_idx * d_dot_n, M<X>::length(d));
                } else {
                    outward_normal = rec.n;
                    ni_over_nt = M<X>::rcp(lf.ref_idx);
                    cosine = M<X>::div(-d_dot_n, M<X>::length(d));
                }
                if (refract<X>(d, outward_normal, ni_over_nt, refracted)) reflect_prob = schlick<X>(cosine, lf.ref_idx);
                else reflect_prob = 1.0;
                nd = (g.next() < reflect_prob) ? reflected : refracted;
                no = rec.p;
                nbeta = beta * d3(1.0, 1.0, 1.0); // specular: albedo * color(next), no emission (trace.rs:209-211)
                done = false;
            } else if (lf.mat_kind == PT_M_METAL) { // src/material.rs:230-241
                D3 reflected = reflect<X>(M<X>::unit_vector(d), rec.n);
                nd = reflected + lf.fuzz * random_in_unit_sphere(g);
                no = rec.p;
                nbeta = beta * d3(lf.albedo[0], lf.albedo[1], lf.albedo[2]);
                done = false;
            } else if (lf.mat_kind == PT_M_LAMBERTIAN || lf.mat_kind == PT_M_ISOTROPIC) {
                const bool lambert = lf.mat_kind == PT_M_LAMBERTIAN;
                // scatter(): src/material.rs:289-301 (Lambertian), 347-351 (Isotropic draws a sphere sample it never uses)
                // With PT_RENDER_BOOK_SEMANTICS that sample is the scattered direction and the pdf is the
                // uniform sphere's, 1/(4 pi) (unit_sphere_pdf, src/material.rs:46-48).
                const bool book = (flags & PT_RENDER_BOOK_SEMANTICS) != 0 && !lambert;
                D3 iso_dir = d3(0.0, 0.0, 0.0);
                if (!lambert) {
                    const D3 sample = random_in_unit_sphere(g);
                    if (book) iso_dir = sample;
                }
                const double iso_pdf = book ? 0.25 * kFrac1Pi : 0.0;
                const D3 albedo = leaf_texture(sv, lf, rec);
                add_radiance(lx, ly, lz, pid, beta * emissive);
#ifndef PT_F64_WEIGHTS
#define PT_F64_WEIGHTS 0
#endif
                if constexpr (X || PT_F64_WEIGHTS) {
                    Onb uvw;
                    if (lambert) uvw = onb_from_w<X>(rec.n);
                    // MixturePDF::generate, src/material.rs:427-433
                    D3 dir;
                    if (g.next() < 0.5) dir = lights_random<X>(sv, g, rec.p);
                    else dir = lambert ? onb_local(uvw, random_cosine_direction<X>(g)) : iso_dir;
                    // MixturePDF::value, src/material.rs:424-426
                    const D3 unit_dir = M<X>::unit_vector(dir);
                    double own = iso_pdf;
                    if (lambert) { // CosinePDF::value, src/material.rs:375-382
                        double cosine = M<X>::dot(unit_dir, uvw.w);
                        own = cosine > 0.0 ? cosine * kFrac1Pi : 0.0;
                    }
                    const double pdf_val = 0.5 * lights_pdf_value<X>(sv, rec.p, dir) + 0.5 * own;
                    double spdf = iso_pdf;
                    if (lambert) { // Lambertian::scattering_pdf, src/material.rs:268-275
                        double cosine = M<X>::dot(rec.n, unit_dir);
                        spdf = cosine < 0.0 ? 0.0 : cosine * kFrac1Pi;
                    }
                    // colour = albedo * spdf * color(next) / pdf_val + emissive (src/trace.rs:217-221)
                    nbeta = M<X>::divs(beta * (albedo * spdf), pdf_val);
                    nd = dir;
                    if constexpr (!X) { // fast mode: see the f32 form below
                        const bool below = lambert && spdf == 0.0;
                        if (below) nbeta = d3(0.0, 0.0, 0.0);
                        dead = below && !sv.has_isotropic;
                    }
                } else {
                    // same estimator with the weights in f32 (see "f32 radiometry" in pt_device.h)
                    const F3 nf = to_f3(rec.n);
                    OnbF uvw;
                    if (lambert) uvw = onbf_from_w(nf);
                    D3 dir;
                    if (g.next() < 0.5) {
                        const double x = (double)sv.n_lights * g.next(); // HitableList::random, hitable.rs:74-77
                        dir = light_random_fast(sv.lights[(int)x], g, rec.p);
                    } else {
                        dir = lambert ? to_d3(onbf_local(uvw, random_cosine_direction_f(g))) : iso_dir;
                    }
                    const F3 dirf = to_f3(dir);
                    const float len2 = dotf(dirf, dirf);
                    const F3 unit_dir = dirf * __builtin_amdgcn_rsqf(len2);
                    float own = (float)iso_pdf, spdf = (float)iso_pdf;
                    if (lambert) {
                        const float cw = dotf(unit_dir, uvw.w);
                        own = cw > 0.0f ? cw * 0.31830988618379067154f : 0.0f;
                        const float cn = dotf(nf, unit_dir);
                        spdf = cn < 0.0f ? 0.0f : cn * 0.31830988618379067154f;
                    }
                    const float weight = __builtin_amdgcn_rcpf((float)sv.n_lights);
                    float lsum = 0.0f;
                    for (int k = 0; k < sv.n_lights; ++k)
                        lsum += weight * light_pdf_value_fast(sv.lights[k], rec.p, dirf, len2);
                    const float pdf_val = 0.5f * lsum + 0.5f * own;
                    const float inv_pdf = __builtin_amdgcn_rcpf(pdf_val);
                    // Isotropic: 0 * x / 0 must stay NaN (src/trace.rs:217-220): rcp(0) = inf, 0 * inf = NaN.
                    // Lambertian: a direction below the surface has scattering_pdf 0 and its sample weighs
                    // nothing. In f64 the mixture pdf is then still positive (the direction came from a
                    // light); in f32 a grazing light sample can also miss its light by rounding, which
                    // would turn 0 / pdf into 0 / 0, so the weight is pinned to the 0 the reference gets.
                    const bool below = lambert && spdf == 0.0f; // a NaN spdf is not "below": it propagates
                    const float wr = below ? 0.0f : ((float)albedo.x * spdf) * inv_pdf;
                    const float wg = below ? 0.0f : ((float)albedo.y * spdf) * inv_pdf;
                    const float wb = below ? 0.0f : ((float)albedo.z * spdf) * inv_pdf;
                    nbeta = beta * d3((double)wr, (double)wg, (double)wb);
                    nd = dir;
                    // A path whose throughput is exactly 0 can only add 0 * (finite radiance) from here on;
                    // without Isotropic materials nothing downstream turns that into NaN, so it stops here.
                    dead = below && !sv.has_isotropic;
                }
                no = rec.p;
                done = false;
            }
        }
        if (done) add_radiance(lx, ly, lz, pid, beta * emissive); // src/trace.rs:225
        alive = !done && !dead;
        rn = g.n;
    }
    if constexpr (FIRST == RAD_FIRST) {
        lx[pid] = first.x;
        ly[pid] = first.y;
        lz[pid] = first.z;
    }
    if constexpr (FIRST == RAD_RETURN) *vertex_radiance = first;
    return alive;
}

// FUSE = 0: hits come from the extend kernel through hit_t / hit_id.
// FUSE = 1 / 2: the scene is a short box-free op list (2: with media); the ray is traced right here before it is
// shaded, so the extend launch, its re-read of the ray and the hit round trip through HBM disappear. The op list is
// passed in sv.ops / sv.n_traversal_ops.
// A leaf record padded to 200 B: with a stride of 50 dwords the same field of different leaves spreads over 16 LDS
// banks instead of 2.
struct LdsLeaf {
    DevLeaf l;
    double pad;
};
constexpr int kLdsLeaves = 32; // the fused kernels keep the whole leaf table in LDS

// LDS of one workgroup. Kept as ONE object: with a second __shared__ object beside a buffer that direct-to-LDS loads
// fill, the compiler waits for those loads before every LDS read.
template <int FUSE> struct ShadeLds {
    uint32_t cnt[kWavesPerBlock + 1];
    LdsLeaf leaves[1];
    unsigned char tile[1][16];
};
template <> struct ShadeLds<1> {
    alignas(16) unsigned char tile[kWavesPerBlock][Queue::kTileBytes]; // each wave's next queue tile
    LdsLeaf leaves[kLdsLeaves];
    uint32_t cnt[kWavesPerBlock + 1];
};

// Starts the copy of queue tile `tile` into the calling wave's LDS buffer: six 1-KiB direct-to-LDS loads (lane l
// moves bytes [16 l, 16 l + 16) of every KiB; the instruction offset advances the global and the LDS address alike).
// Nothing waits here. The loads are written in assembly on purpose: told about them, the compiler drains the VM
// counter before EVERY later LDS read (it cannot tell the tile buffer from the leaf table), which would put the
// whole HBM round trip back in front of the shading. The caller retires them with tile_wait() before it reads the
// buffer.
__device__ __forceinline__ void prefetch_tile(const Queue &q, uint32_t tile, unsigned char *lds_wave, uint32_t lane) {
    typedef __attribute__((address_space(3))) unsigned char *LPtr;
    const char *g0 = q.base + (size_t)tile * Queue::kTileBytes + lane * 16u;
    const char *g1 = g0 + 4096;
    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(LPtr)lds_wave);
    const uint32_t l1 = l0 + 4096u;
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, off offset:3072\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 :
                 : "v"(g0), "v"(g1), "s"(l0), "s"(l1)
                 : "memory", "m0");
}
// Every vector memory operation of the wave, the tile copy included, has completed.
__device__ __forceinline__ void tile_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <bool X, bool NEED_UV, int FUSE>
__global__ __launch_bounds__(kBlock, X ? 1 : PT_SHADE_WAVES) void k_shade(Queue qi, Queue qo, QueueShape sh, const uint32_t *n_in,
                                                  uint32_t *n_out, SceneView sv, const double *hit_t, const int32_t *hit_id,
                                                  double *lx, double *ly, double *lz,
                                                  uint32_t depth, uint32_t max_depth, uint32_t flags) {
    __shared__ ShadeLds<(FUSE != 0) ? 1 : 0> S;
    uint32_t *lds = S.cnt;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_virtual = ((max_shard_count(n_in, sh.shards) + kBlock - 1) / kBlock) << sh.shift;
    // first queue tile of virtual chunk v that this wave handles
    auto tile_of = [&](uint32_t v) -> uint32_t {
        return (((v & (sh.shards - 1)) * sh.shard_cap + (v >> sh.shift) * kBlock) >> 6) + wave;
    };
    if constexpr (FUSE != 0) {
        if (blockIdx.x < n_virtual) prefetch_tile(qi, tile_of(blockIdx.x), S.tile[wave], lane);
        // Shading gathers ~100 B of its leaf per lane through a chain of dependent loads (kind -> geometry ->
        // material); from LDS each link costs tens of cycles instead of an L1/L2 round trip.
        constexpr int kWords = (int)(sizeof(DevLeaf) / sizeof(double));
        const double *src = (const double *)sv.leaves;
        for (int k = threadIdx.x; k < sv.n_leaves * kWords; k += kBlock)
            ((double *)&S.leaves[k / kWords].l)[k % kWords] = src[k];
        tile_wait();
        __syncthreads();
    }
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (sh.shards - 1);
        const uint32_t ns = shard_count(n_in, s_id);
        const uint32_t row0 = (v >> sh.shift) * kBlock;
        bool alive = false;
        D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0);
        double time = 0.0;
        uint32_t pid = 0, rn = 0;
        D3 o, d, beta;
        double ht = 0.0;
        int32_t id = -1;
        uint64_t pkey = 0;
        if constexpr (FUSE != 0) {
            // This wave's tile was requested one iteration ago (or before the loop) and has had a whole trace + shade
            // to arrive. Every field moves into registers, then the buffer is handed to the request for the tile of
            // the next iteration, which the work below hides. Lanes past the shard's fill read stale entries: they
            // are traced (whole waves walk the op list together) and dropped.
            const unsigned char *e = S.tile[wave] + lane * 8u;
            o = d3(*(const double *)(e + 0 * 512), *(const double *)(e + 1 * 512), *(const double *)(e + 2 * 512));
            d = d3(*(const double *)(e + 3 * 512), *(const double *)(e + 4 * 512), *(const double *)(e + 5 * 512));
            time = *(const double *)(e + 6 * 512);
            beta = d3(*(const double *)(e + 7 * 512), *(const double *)(e + 8 * 512), *(const double *)(e + 9 * 512));
            pkey = *(const uint64_t *)(e + 10 * 512);
            pid = *(const uint32_t *)(S.tile[wave] + 5632 + lane * 4u);
            rn = *(const uint32_t *)(S.tile[wave] + 5888 + lane * 4u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads are done before the buffer is refilled
            const uint32_t vn = v + gridDim.x;
            if (vn < n_virtual) prefetch_tile(qi, tile_of(vn), S.tile[wave], lane);
        }
        if (row0 >= ns) { // uniform for the workgroup
            if constexpr (FUSE != 0) tile_wait();
            continue;
        }
        const uint32_t local = row0 + threadIdx.x;
        const uint32_t i = s_id * sh.shard_cap + local;
        const uint32_t n = s_id * sh.shard_cap + ns; // first slot past this shard's entries
        if (FUSE) {
            extend_list<FUSE == 2>(sv, sv.ops, o, d, time, pkey, depth, 0.001, kF64Max, ht, id);
#ifdef PT_EXPERIMENT_MISS
            id = ht > 1e300 ? id : -1;
#endif
        }
        if (i < n) {
            if (!FUSE) {
                o = d3(qi.ox(i), qi.oy(i), qi.oz(i));
                d = d3(qi.dx(i), qi.dy(i), qi.dz(i));
                time = qi.tm(i);
                pid = qi.pid(i);
                id = hit_id[i];
                ht = hit_t[i];
                beta = d3(qi.bx(i), qi.by(i), qi.bz(i));
                rn = qi.rn(i);
            }
            if (!FUSE) pkey = qi.key(i);
            const DevLeaf *lfp;
            const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
            if constexpr (FUSE != 0) lfp = &S.leaves[li].l;
            else lfp = &sv.leaves[li];
            alive = shade_vertex<X, NEED_UV>(sv, *lfp, o, d, time, beta, pkey, pid, rn, ht, id, depth, max_depth, flags,
                                             lx, ly, lz, no, nd, nbeta);
        }
        const uint32_t slot = s_id * sh.shard_cap + compact_slot(alive, &n_out[s_id], lds);
        // The next tile was requested a whole trace + shade ago; retiring it here, ahead of this iteration's stores,
        // costs next to nothing, while a wait at the top of the loop would also sit out those stores.
        if constexpr (FUSE != 0) tile_wait();
        if (alive) {
            qo.ox(slot) = no.x; qo.oy(slot) = no.y; qo.oz(slot) = no.z;
            qo.dx(slot) = nd.x; qo.dy(slot) = nd.y; qo.dz(slot) = nd.z;
            qo.tm(slot) = time;
            qo.bx(slot) = nbeta.x; qo.by(slot) = nbeta.y; qo.bz(slot) = nbeta.z;
            qo.pid(slot) = pid;
            qo.rn(slot) = rn;
            qo.key(slot) = pkey;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Bounce 0 of an offline pass of a fused scene: the camera sample is generated, traced and shaded in one go, so the
// first queue (96 B per path written by raygen and read back by the first shade launch) never exists. Same chunk
// to shard mapping as k_raygen, same camera arithmetic, same draws.
// ------------------------------------------------------------------------------------------------------------
template <bool NEED_UV, int FUSE>
__global__ __launch_bounds__(kBlock, PT_SHADE_WAVES) void k_first_bounce(Queue qo, QueueShape sh, uint32_t *counts,
                                                                         SceneView sv, CameraDev cam, PassParams pp,
                                                                         double *lx, double *ly, double *lz,
                                                                         uint32_t flags) {
    __shared__ ShadeLds<1> S;
    {
        constexpr int kWords = (int)(sizeof(DevLeaf) / sizeof(double));
        const double *src = (const double *)sv.leaves;
        for (int k = threadIdx.x; k < sv.n_leaves * kWords; k += kBlock)
            ((double *)&S.leaves[k / kWords].l)[k % kWords] = src[k];
        __syncthreads();
    }
    uint32_t *n_out = counts + kShards;
    const uint32_t n = pp.cpix * pp.s_count;
    const uint32_t n_chunks = (n + kBlock - 1) / kBlock;
    for (uint32_t j = blockIdx.x; j < n_chunks; j += gridDim.x) {
        const uint32_t s_id = j & (sh.shards - 1);
        const uint32_t pid = j * kBlock + threadIdx.x;
        const bool valid = pid < n;
        const uint32_t pc = valid ? pid : 0u; // lanes past the end trace path 0 again and are dropped
        const uint32_t p = pc % pp.cpix, s = pc / pp.cpix;
        const uint32_t i = pp.cx0 + p % pp.cw;
        const uint32_t jy = pp.cy0 + (p / pp.cw) * pp.row_step;
        const uint64_t pixel = (uint64_t)jy * pp.width + i;
        Rng g{path_key(pp.seed, pp.frame, pixel, pp.s_base + s), 0};
        double random = g.next(); // src/trace.rs:153-156
        const double u = ((double)i + random) / (double)pp.width;
        random = g.next();
        const double v = ((double)jy + random) / (double)pp.height;
        const D3 rd = random_in_unit_disk(g) * cam.lens_radius; // Camera::get_ray, src/camera.rs:53-59
        const D3 offset = cam.u * rd.x + cam.v * rd.y;
        const double time = cam.time0 + g.next() * (cam.time1 - cam.time0);
        const D3 d = cam.lower_left_corner + cam.horizontal * u + cam.vertical * v - cam.origin - offset;
        const D3 o = cam.origin + offset;
        const D3 beta = d3(1.0, 1.0, 1.0);
        uint32_t rn = g.n;
        double ht;
        int32_t id;
        extend_list<FUSE == 2>(sv, sv.ops, o, d, time, g.key, 0u, 0.001, kF64Max, ht, id);
        bool alive = false;
        D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0);
        if (valid) {
            const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
            alive = shade_vertex<false, NEED_UV, RAD_FIRST>(sv, S.leaves[li].l, o, d, time, beta, g.key, pid, rn, ht, id, 0u,
                                                       pp.max_depth, flags, lx, ly, lz, no, nd, nbeta);
        }
        const uint32_t slot = s_id * sh.shard_cap + compact_slot(alive, &n_out[s_id], S.cnt);
        if (alive) {
            qo.ox(slot) = no.x; qo.oy(slot) = no.y; qo.oz(slot) = no.z;
            qo.dx(slot) = nd.x; qo.dy(slot) = nd.y; qo.dz(slot) = nd.z;
            qo.tm(slot) = time;
            qo.bx(slot) = nbeta.x; qo.by(slot) = nbeta.y; qo.bz(slot) = nbeta.z;
            qo.pid(slot) = pid;
            qo.rn(slot) = rn;
            qo.key(slot) = g.key;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < sh.shards) {
        // segments of bounce 0, per shard, as k_raygen counts them (statistics only: nothing reads a first queue)
        const uint32_t k = threadIdx.x;
        uint32_t fill = 0;
        if (n_chunks > k) {
            const uint32_t mine = (n_chunks - k + sh.shards - 1) >> sh.shift;
            fill = mine * kBlock;
            const uint32_t last = k + (mine - 1) * sh.shards;
            if (last == n_chunks - 1) fill -= n_chunks * kBlock - n;
        }
        counts[k] = fill;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form of the fused pipeline: ONE launch traces a whole pass. The pass's paths are dealt to the resident
// waves in contiguous runs of queue tiles; a wave then carries its own paths through every bounce, compacting the
// survivors into the same run of tiles of the other queue. A wave shares nothing with any other wave: no atomics on
// shared counters, no barriers, no launch per bounce, and the tile prefetch runs on across bounces. The price is one
// partly filled tile per wave and bounce, a few per cent of the tile steps. Which wave carries a path has no effect
// on the path (its RNG is keyed by pixel and sample), so images are bit-identical to the launch-per-bounce form.
// ------------------------------------------------------------------------------------------------------------
template <bool NEED_UV, int FUSE>
__global__ __launch_bounds__(kBlock, PT_SHADE_WAVES) void k_trace_persistent(Queue qa, Queue qb, uint32_t *counts,
                                                                             SceneView sv, double *lx, double *ly,
                                                                             double *lz, uint32_t max_depth,
                                                                             uint32_t flags) {
    __shared__ ShadeLds<1> S;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    {
        constexpr int kWords = (int)(sizeof(DevLeaf) / sizeof(double));
        const double *src = (const double *)sv.leaves;
        for (int k = threadIdx.x; k < sv.n_leaves * kWords; k += kBlock)
            ((double *)&S.leaves[k / kWords].l)[k % kWords] = src[k];
        __syncthreads(); // the only barrier of the kernel
    }
    // this wave's run of tiles: the same in both queues
    const uint32_t n_paths = shard_count(counts, 0); // raygen's fill of the first queue
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    const uint32_t n_tiles = (n_paths + 63u) >> 6;
    const uint32_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const uint32_t first_tile = (blockIdx.x * kWavesPerBlock + wave) * per_wave;
    uint32_t n = 0; // entries in this wave's run
    if (first_tile < n_tiles) n = min(per_wave * 64u, n_paths - first_tile * 64u);
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    Queue qi = qa, qo = qb;
    for (uint32_t depth = 0; depth <= max_depth && n > 0; ++depth) {
        const uint32_t tiles = (n + 63u) >> 6;
        uint32_t n_next = 0;
        prefetch_tile(qi, first_tile, S.tile[wave], lane);
        tile_wait();
        for (uint32_t t = 0; t < tiles; ++t) {
            const unsigned char *e = S.tile[wave] + lane * 8u;
            const D3 o = d3(*(const double *)(e + 0 * 512), *(const double *)(e + 1 * 512), *(const double *)(e + 2 * 512));
            const D3 d = d3(*(const double *)(e + 3 * 512), *(const double *)(e + 4 * 512), *(const double *)(e + 5 * 512));
            const double time = *(const double *)(e + 6 * 512);
            const D3 beta = d3(*(const double *)(e + 7 * 512), *(const double *)(e + 8 * 512), *(const double *)(e + 9 * 512));
            const uint64_t pkey = *(const uint64_t *)(e + 10 * 512);
            const uint32_t pid = *(const uint32_t *)(S.tile[wave] + 5632 + lane * 4u);
            uint32_t rn = *(const uint32_t *)(S.tile[wave] + 5888 + lane * 4u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads are done before the buffer is refilled
            if (t + 1 < tiles) prefetch_tile(qi, first_tile + t + 1, S.tile[wave], lane);
            double ht;
            int32_t id;
            extend_list<FUSE == 2>(sv, sv.ops, o, d, time, pkey, depth, 0.001, kF64Max, ht, id);
            bool alive = false;
            D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0);
            if (t * 64u + lane < n) { // lanes past the fill traced a stale entry; they are dropped here
                const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
                alive = shade_vertex<false, NEED_UV>(sv, S.leaves[li].l, o, d, time, beta, pkey, pid, rn, ht, id, depth,
                                                     max_depth, flags, lx, ly, lz, no, nd, nbeta);
            }
            const unsigned long long ballot = __ballot(alive);
            const uint32_t slot = first_tile * 64u + n_next + __popcll(ballot & ((1ull << lane) - 1ull));
            n_next += (uint32_t)__popcll(ballot);
            tile_wait(); // the next tile has landed; see k_shade
            if (alive) {
                qo.ox(slot) = no.x; qo.oy(slot) = no.y; qo.oz(slot) = no.z;
                qo.dx(slot) = nd.x; qo.dy(slot) = nd.y; qo.dz(slot) = nd.z;
                qo.tm(slot) = time;
                qo.bx(slot) = nbeta.x; qo.by(slot) = nbeta.y; qo.bz(slot) = nbeta.z;
                qo.pid(slot) = pid;
                qo.rn(slot) = rn;
                qo.key(slot) = pkey;
            }
        }
        tile_wait(); // this bounce's entries are written before the next bounce reads them back
        // segments of the next bounce, for the statistics (bounce 0 was counted by raygen)
        if (lane == 0 && n_next) atomicAdd(&counts[(size_t)(depth + 1) * kShards], n_next);
        const Queue tmp = qi;
        qi = qo;
        qo = tmp;
        n = n_next;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Path pool form of the fused pipeline: ONE launch traces a whole offline pass and no path ever visits HBM. A wave is
// a pool of 64 path slots over its own run of the pass's path ids: every round all 64 lanes walk the primitive list,
// the lanes that hold a path shade their vertex and keep the next ray, throughput and radiance sum in registers; a
// path that ends stores its radiance, and once enough slots are free the free lanes generate the next camera samples
// of the run (ballot + prefix count; the run belongs to this wave alone). Queue entries, their compaction and the
// launch per bounce are gone; what crosses HBM per PATH is the 24-B radiance the resolve kernel reads. Same camera
// arithmetic and draws as k_raygen, same vertex code, same order of the per-path sum: images are bit-identical to
// the other forms.
// ------------------------------------------------------------------------------------------------------------
// Camera and pass constants of the pool kernel, read from memory where camera samples are made: as kernel arguments
// they would sit in some sixty scalar registers for the whole kernel, which the walk and the shading need themselves.
struct PoolConsts {
    CameraDev cam;
    PassParams pp;
};
struct PoolLds {
    LdsLeaf leaves[kLdsLeaves];
    // camera samples made 64 at a time by the whole wave and handed to lanes as they fall free
    double cam_ray[kWavesPerBlock][7][64]; // origin, direction, time
    uint64_t cam_key[kWavesPerBlock][64];
    uint32_t cam_rn[kWavesPerBlock][64];
    uint32_t cam_pid[kWavesPerBlock][64];
};
#ifndef PT_EXACT_POOL_WAVES
#define PT_EXACT_POOL_WAVES 3 // measured on the Cornell box: 1 -> 895, 2 -> 1506, 3 -> 1786, 4 -> 1704 Mpaths/s
#endif
// X: exact mode. The walk is the wave-uniform scan of the reference's own threaded tree (every box test, the
// reference's order), the arithmetic is M<true>'s; sv.ops is then the full stream, not the fused list.
template <bool NEED_UV, int FUSE, bool X = false>
__global__ __launch_bounds__(kBlock, X ? PT_EXACT_POOL_WAVES : PT_SHADE_WAVES) void k_path_pool(SceneView sv, const PoolConsts *consts, uint32_t n,
                                                                      uint32_t max_depth, double *lx, double *ly,
                                                                      double *lz, uint32_t *counts, uint32_t flags,
                                                                      uint32_t refill_at, uint32_t chunk) {
    __shared__ PoolLds S;
    {
        constexpr int kWords = (int)(sizeof(DevLeaf) / sizeof(double));
        const double *src = (const double *)sv.leaves;
        for (int k = threadIdx.x; k < sv.n_leaves * kWords; k += kBlock)
            ((double *)&S.leaves[k / kWords].l)[k % kWords] = src[k];
        __syncthreads(); // the only barrier of the kernel
    }
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    // Path ids are handed out in chunks from one counter (the first slot past the bounce rows of counts[], zero at
    // launch and left out of the segment sum): a wave takes the next chunk when its current one is used up, so every
    // wave stays full until the pass itself runs out of paths.
    uint32_t cur = 0, end = 0, stash_n = 0;
    bool exhausted = false;
    bool has = false;
    D3 o = d3(0.0, 0.0, 0.0), d = d3(0.0, 0.0, 1.0), beta = d3(0.0, 0.0, 0.0), sum = d3(0.0, 0.0, 0.0);
    double time = 0.0;
    uint64_t pkey = 0;
    uint32_t pid = 0, rn = 0, depth = 0, segments = 0;
    for (;;) {
        bool drained = false;
        for (;;) { // hand camera samples to free lanes
            const unsigned long long idle = __ballot(!has);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle < refill_at) break;
            if (stash_n == 0) {
                if (cur >= end && !exhausted) {
                    uint32_t first = 0;
                    if (lane == 0) first = atomicAdd(&counts[(size_t)(max_depth + 1) * kShards], chunk);
                    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                    exhausted = first >= n;
                    cur = min(first, n);
                    end = min(cur + chunk, n); // chunk is far below 2^32 - n
                }
                if (cur >= end) { // the pass has no more paths to start
                    drained = n_idle == 64;
                    break;
                }
                // the next (up to) 64 camera samples, one per lane, parked in LDS
                const PoolConsts *pc = consts;
                asm volatile("" : "+s"(pc)); // keeps the loads below in this block
                const CameraDev cam = pc->cam;
                const PassParams pp = pc->pp;
                const uint32_t m = min(64u, end - cur);
                const bool fresh = lane < m;
                const uint32_t id = cur + lane;
                uint32_t i = 0, jy = 0, smp = 0;
                uint64_t pixel = 0;
                if (fresh) {
                    const uint32_t p = id % pp.cpix;
                    smp = id / pp.cpix;
                    i = pp.cx0 + p % pp.cw;
                    jy = pp.cy0 + (p / pp.cw) * pp.row_step;
                    pixel = (uint64_t)jy * pp.width + i;
                }
                const unsigned long long made = __ballot(fresh);
                if (fresh) {
                    const uint32_t k = (uint32_t)__popcll(made & ((1ull << lane) - 1ull));
                    Rng g{path_key(pp.seed, pp.frame, pixel, pp.s_base + smp), 0};
                    double random = g.next(); // src/trace.rs:153-156
                    const double u = ((double)i + random) / (double)pp.width;
                    random = g.next();
                    const double v = ((double)jy + random) / (double)pp.height;
                    const D3 rd = random_in_unit_disk(g) * cam.lens_radius; // Camera::get_ray, src/camera.rs:53-59
                    const D3 offset = cam.u * rd.x + cam.v * rd.y;
                    const double t = cam.time0 + g.next() * (cam.time1 - cam.time0);
                    const D3 dir = cam.lower_left_corner + cam.horizontal * u + cam.vertical * v - cam.origin - offset;
                    const D3 org = cam.origin + offset;
                    S.cam_ray[wv][0][k] = org.x; S.cam_ray[wv][1][k] = org.y; S.cam_ray[wv][2][k] = org.z;
                    S.cam_ray[wv][3][k] = dir.x; S.cam_ray[wv][4][k] = dir.y; S.cam_ray[wv][5][k] = dir.z;
                    S.cam_ray[wv][6][k] = t;
                    S.cam_key[wv][k] = g.key;
                    S.cam_rn[wv][k] = g.n;
                    S.cam_pid[wv][k] = id;
                }
                stash_n = (uint32_t)__popcll(made);
                cur += m;
            }
            const uint32_t take = min(n_idle, stash_n);
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!has && rank < take) {
                const uint32_t k = stash_n - 1u - rank;
                pid = S.cam_pid[wv][k];
                o = d3(S.cam_ray[wv][0][k], S.cam_ray[wv][1][k], S.cam_ray[wv][2][k]);
                d = d3(S.cam_ray[wv][3][k], S.cam_ray[wv][4][k], S.cam_ray[wv][5][k]);
                time = S.cam_ray[wv][6][k];
                pkey = S.cam_key[wv][k];
                rn = S.cam_rn[wv][k];
                beta = d3(1.0, 1.0, 1.0);
                depth = 0;
                has = true;
            }
            stash_n -= take;
        }
        if (drained) break; // nothing in flight, nothing left to start
        segments += (uint32_t)__popcll(__ballot(has));
        double ht;
        int32_t id;
        if constexpr (X) {
            unsigned long long unused_nodes = 0, unused_prims = 0;
            extend_scan_walk<true, true, FUSE == 2, false>(sv, o, d, time, pkey, depth, 0.001, kF64Max, has, false, ht, id,
                                                           unused_nodes, unused_prims);
        } else {
            extend_list<FUSE == 2>(sv, sv.ops, o, d, time, pkey, depth, 0.001, kF64Max, ht, id); // free lanes walk along
        }
        if (has) {
            const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
            D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0), c;
            const bool alive = shade_vertex<X, NEED_UV, RAD_RETURN>(sv, S.leaves[li].l, o, d, time, beta, pkey, pid, rn, ht,
                                                                        id, depth, max_depth, flags, lx, ly, lz, no, nd,
                                                                        nbeta, &c);
            // the per-path sum of the other forms: the first vertex is stored, later ones are added unless zero
            if (depth == 0) sum = c;
            else if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) sum = sum + c;
            if (alive) {
                o = no;
                d = nd;
                beta = nbeta;
                depth += 1;
            } else {
                lx[pid] = sum.x;
                ly[pid] = sum.y;
                lz[pid] = sum.z;
                has = false;
            }
        }
    }
    if (lane == 0 && segments) atomicAdd(&counts[0], segments); // the pass's segment count, for the statistics
}

// ------------------------------------------------------------------------------------------------------------
// The path pool for scenes that are traced through the SAH tree: one launch per offline pass, a path in one lane's
// registers from camera sample to last vertex, no queues. A lane is free, on an inner node, on a leaf, or waiting
// with a finished traversal. Every round steps the traversing lanes (inner nodes first, then leaves, as in
// k_extend_accel_pool), and once enough lanes wait (or too few still traverse) the waiting lanes shade their vertex
// together and start their next ray; free lanes take new camera samples from the pass's chunk counter.
// ------------------------------------------------------------------------------------------------------------
#ifndef PT_TREE_UV_WAVES
#define PT_TREE_UV_WAVES 2 // waves per SIMD of the textured variants: their shading is the largest code here (4: 638, 3: 640, 2: 730 Mpaths/s on the final scene)
#endif
template <bool NEED_UV, bool MEDIA>
__global__ __launch_bounds__(kBlock, NEED_UV ? PT_TREE_UV_WAVES : PT_SHADE_WAVES) void k_tree_pool(SceneView sv, const AccelNode *nodes, const DevOp *prims,
                                                                      const PoolConsts *consts, uint32_t n, uint32_t max_depth,
                                                                      double *lx, double *ly, double *lz, uint32_t *counts,
                                                                      uint32_t flags, uint32_t tuning, uint32_t chunk) {
    __shared__ int32_t stack[kAccelMaxDepth][kBlock]; // [depth][thread]: conflict-free
    const uint32_t refill_at = tuning & 0xffu, inner_min = (tuning >> 8) & 0xffu, shade_min = (tuning >> 16) & 0xffu,
                   trav_min = tuning >> 24;
    constexpr int32_t kDone = INT32_MIN; // never a leaf reference
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t cur = 0, end = 0;
    bool exhausted = false;
    bool has = false;
    D3 o = d3(0, 0, 0), d = d3(0, 0, 1), inv = o, beta = o, sum = o;
    double time = 0.0, closest = 0.0;
    uint64_t pkey = 0;
    uint32_t pid = 0, rn = 0, depth = 0, segments = 0;
    int32_t best = -1, node = kDone;
    int sp = 0;
    auto start_ray = [&]() { // (o, d) hold a new ray
        inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
        closest = kF64Max;
        best = -1;
        sp = 0;
        // rays with a NaN or infinite component are not traced (see k_extend_accel)
        const double probe = (o.x + o.y + o.z) + (d.x + d.y + d.z);
        node = (probe == probe && fabs(probe) <= kF64Max) ? 0 : kDone;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle >= refill_at) {
            if (cur >= end && !exhausted) {
                uint32_t first = 0;
                if (lane == 0) first = atomicAdd(&counts[(size_t)(max_depth + 1) * kShards], chunk);
                first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                exhausted = first >= n;
                cur = min(first, n);
                end = min(cur + chunk, n);
            }
            if (cur < end) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!has && cur + rank < end) {
                    const PoolConsts *pc = consts;
                    asm volatile("" : "+s"(pc)); // keeps the loads below in this block
                    const CameraDev cam = pc->cam;
                    const PassParams pp = pc->pp;
                    pid = cur + rank;
                    const uint32_t p = pid % pp.cpix, s = pid / pp.cpix;
                    const uint32_t i = pp.cx0 + p % pp.cw;
                    const uint32_t jy = pp.cy0 + (p / pp.cw) * pp.row_step;
                    const uint64_t pixel = (uint64_t)jy * pp.width + i;
                    Rng g{path_key(pp.seed, pp.frame, pixel, pp.s_base + s), 0};
                    double random = g.next(); // src/trace.rs:153-156
                    const double u = ((double)i + random) / (double)pp.width;
                    random = g.next();
                    const double v = ((double)jy + random) / (double)pp.height;
                    const D3 rd = random_in_unit_disk(g) * cam.lens_radius; // Camera::get_ray, src/camera.rs:53-59
                    const D3 offset = cam.u * rd.x + cam.v * rd.y;
                    time = cam.time0 + g.next() * (cam.time1 - cam.time0);
                    d = cam.lower_left_corner + cam.horizontal * u + cam.vertical * v - cam.origin - offset;
                    o = cam.origin + offset;
                    beta = d3(1.0, 1.0, 1.0);
                    pkey = g.key;
                    rn = g.n;
                    depth = 0;
                    has = true;
                    start_ray();
                }
                cur = min(cur + n_idle, end);
            } else if (n_idle == 64) {
                break; // nothing in flight, nothing left to start
            }
        }
        // traversal: inner nodes for as long as at least `inner_min` lanes are on one, then the leaves
        for (;;) {
            if (has && node >= 0) {
                const AccelNode &nd = nodes[node];
                double t0, t1;
                const bool h0 = accel_box(nd.lo[0], nd.hi[0], o, inv, 0.001, closest, t0);
                const bool h1 = accel_box(nd.lo[1], nd.hi[1], o, inv, 0.001, closest, t1);
                const int32_t c0 = nd.child[0], c1 = nd.child[1];
                if (h0 && h1) {
                    const bool first0 = t0 <= t1;
                    stack[sp++][threadIdx.x] = first0 ? c1 : c0;
                    node = first0 ? c0 : c1;
                } else if (h0 || h1) {
                    node = h0 ? c0 : c1;
                } else {
                    node = sp ? stack[--sp][threadIdx.x] : kDone;
                }
            }
            const uint32_t n_inner = (uint32_t)__popcll(__ballot(has && node >= 0));
            if (n_inner == 0 || n_inner < inner_min) break;
        }
        if (has && node < 0 && node != kDone) {
            const int32_t ref = ~node;
            const int32_t first = ref >> 3, count = ref & 7;
            for (int32_t k = first; k < first + count; ++k) {
                const DevOp op = prims[k];
                double t = 0.0;
                int sub = 0;
                bool hit;
                if (MEDIA && (op.kind & 0xff) == OP_MEDIUM)
                    hit = medium_hit<false>(sv, op, o, d, inv, time, 0.001, closest, pkey, depth, t);
                else hit = prim_hit<false>(sv, op, o, d, inv, time, 0.001, closest, t, sub);
                if (hit) {
                    closest = t;
                    best = op.leaf | (sub << 28);
                }
            }
            node = sp ? stack[--sp][threadIdx.x] : kDone;
        }
        // vertices: when enough lanes wait with a finished traversal, or too few are still traversing to go on
        const bool ready = has && node == kDone;
        const uint32_t n_ready = (uint32_t)__popcll(__ballot(ready));
        const uint32_t n_trav = (uint32_t)__popcll(__ballot(has && node != kDone));
        if (n_ready && (n_ready >= shade_min || n_trav <= trav_min)) {
            segments += n_ready;
            if (ready) {
                const int32_t li = best < 0 ? 0 : (best & 0x0fffffff);
                D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0), c;
                const bool alive = shade_vertex<false, NEED_UV, RAD_RETURN>(sv, sv.leaves[li], o, d, time, beta, pkey, pid, rn,
                                                                            closest, best, depth, max_depth, flags, lx, ly,
                                                                            lz, no, nd, nbeta, &c);
                if (depth == 0) sum = c; // the per-path sum of the other forms
                else if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) sum = sum + c;
                if (alive) {
                    o = no;
                    d = nd;
                    beta = nbeta;
                    depth += 1;
                    start_ray();
                } else {
                    lx[pid] = sum.x;
                    ly[pid] = sum.y;
                    lz[pid] = sum.z;
                    has = false;
                }
            }
        }
    }
    if (lane == 0 && segments) atomicAdd(&counts[0], segments);
}

// ------------------------------------------------------------------------------------------------------------
// resolve: per-pixel sum in sample order, then the write of src/trace.rs:166-182
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_resolve(const double *lx, const double *ly, const double *lz, double *ax,
                                                    double *ay, double *az, float *image, uint32_t *frames_per_pixel,
                                                    PassParams pp, const uint32_t *counts, unsigned long long *stats) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < pp.cpix; p += gridDim.x * kBlock) {
        double sx = pp.first_pass ? 0.0 : ax[p];
        double sy = pp.first_pass ? 0.0 : ay[p];
        double sz = pp.first_pass ? 0.0 : az[p];
        for (uint32_t s = 0; s < pp.s_count; ++s) { // pixel_colour += color(...) in sample order
            const uint32_t pid = s * pp.cpix + p;
            sx += lx[pid];
            sy += ly[pid];
            sz += lz[pid];
        }
        if (!pp.last_pass) {
            ax[p] = sx; ay[p] = sy; az[p] = sz;
            continue;
        }
        const uint32_t i = pp.cx0 + p % pp.cw;
        const uint32_t j = pp.cy0 + (p / pp.cw) * pp.row_step;
        const uint64_t pixel = (uint64_t)j * pp.width + i;
        float *px = image + pixel * 4;
        const double ns = (double)pp.spp;
        const double cx = sx / ns, cy = sy / ns, cz = sz / ns; // src/trace.rs:166
        if (pp.mode == PT_MODE_REALTIME) {
            if (pixel_skip_draw(pp.seed, pp.frame, pixel) < 0.8) continue; // src/trace.rs:144-146
            uint32_t c = frames_per_pixel[pixel];
            c += (c <= 1000u) ? 1u : 0u; // src/trace.rs:149
            frames_per_pixel[pixel] = c;
            const float weight = 1.0f / (float)c; // src/trace.rs:171-177
            const float one_minus_weight = 1.0f - weight;
            px[0] = (float)cx * weight + px[0] * one_minus_weight;
            px[1] = (float)cy * weight + px[1] * one_minus_weight;
            px[2] = (float)cz * weight + px[2] * one_minus_weight;
        } else {
            px[0] = (float)cx; // src/trace.rs:179-181
            px[1] = (float)cy;
            px[2] = (float)cz;
        }
        if (px[0] != px[0] || px[1] != px[1] || px[2] != px[2]) atomicAdd(&stats[1], 1ull);
    }
    if (blockIdx.x == 0) { // segments traced in this pass = sum of every bounce's queue fill
        unsigned long long seg = 0;
        const uint32_t total = (pp.max_depth + 1) * kShards;
        for (uint32_t k = threadIdx.x; k < total; k += kBlock) seg += counts[k];
        if (seg) atomicAdd(&stats[0], seg);
    }
}

// ------------------------------------------------------------------------------------------------------------
// post kernels: PPM conversion and the display transform
// ------------------------------------------------------------------------------------------------------------
// reinhard_tonemap (src/trace.rs:236-242) + sqrt gamma + `as u8` (src/lib.rs:330-340), rows flipped as the PPM
// writer does (src/lib.rs:617-630). f64 with IEEE division and sqrt, as on the host.
__global__ __launch_bounds__(kBlock) void k_tonemap_rgb8(const float *image, uint32_t width, uint32_t height,
                                                         uint8_t *out) {
    const uint32_t n = width * height;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const uint32_t i = p % width, j = p / width;
        const float *px = image + ((size_t)(height - j - 1) * width + i) * 4;
        for (int c = 0; c < 3; ++c) {
            const double colour = (double)px[c] * 1.5;
            const double mapped = colour / (colour + 1.0);
            const double scaled = 255.99 * sqrt(mapped);
            uint8_t b; // Rust's `as u8` saturates and maps NaN to 0
            if (!(scaled > 0.0)) b = 0;
            else if (scaled >= 255.0) b = 255;
            else b = (uint8_t)scaled;
            out[(size_t)p * 3 + c] = b;
        }
    }
}
// assets/shaders/tonemap.frag:37-70, f32 like the shader. `v * M` with a column-major mat3 is a dot product of v
// with each constructor triple.
__device__ __forceinline__ float aces_fit(float v) { // RRTAndODTFit, tonemap.frag:37-42
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
__global__ __launch_bounds__(kBlock) void k_display_transform(const float *image, uint32_t width, uint32_t height,
                                                              float exposure, float *out) {
    const uint32_t n = width * height;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const uint32_t i = p % width, j = p / width; // j = 0 is the top of the screen
        const float *px = image + ((size_t)(height - j - 1) * width + i) * 4;
        const float r = px[0] * exposure, g = px[1] * exposure, b = px[2] * exposure;
        const float ir = r * 0.59719f + g * 0.35458f + b * 0.04823f;
        const float ig = r * 0.07600f + g * 0.90834f + b * 0.01566f;
        const float ib = r * 0.02840f + g * 0.13383f + b * 0.83777f;
        const float fr = aces_fit(ir), fg = aces_fit(ig), fb = aces_fit(ib);
        float orr = fr * 1.60475f + fg * -0.53108f + fb * -0.07367f;
        float org = fr * -0.10208f + fg * 1.10813f + fb * -0.00605f;
        float orb = fr * -0.00327f + fg * -0.07276f + fb * 1.07602f;
        orr = fminf(fmaxf(orr, 0.0f), 1.0f);
        org = fminf(fmaxf(org, 0.0f), 1.0f);
        orb = fminf(fmaxf(orb, 0.0f), 1.0f);
        // uv of the fragment centre after the shader's flip; the distance is the same either way up
        const float u = ((float)i + 0.5f) / (float)width, v = 1.0f - ((float)j + 0.5f) / (float)height;
        const float du = u - 0.5f, dv = v - 0.5f;
        const float vignette = 1.0f - sqrtf(du * du + dv * dv);
        float *o = out + (size_t)p * 4;
        o[0] = orr * vignette;
        o[1] = org * vignette;
        o[2] = orb * vignette;
        o[3] = 1.0f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// pt_intersect helpers
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_load_rays(const pt_ray *rays, uint32_t n, Queue q) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const pt_ray r = rays[i];
        q.ox(i) = r.origin[0]; q.oy(i) = r.origin[1]; q.oz(i) = r.origin[2];
        q.dx(i) = r.direction[0]; q.dy(i) = r.direction[1]; q.dz(i) = r.direction[2];
        q.tm(i) = r.time;
    }
}
__global__ __launch_bounds__(kBlock) void k_load_rays_soa(const double *soa, uint32_t n, Queue q) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        q.ox(i) = soa[i]; q.oy(i) = soa[(size_t)n + i]; q.oz(i) = soa[2 * (size_t)n + i];
        q.dx(i) = soa[3 * (size_t)n + i]; q.dy(i) = soa[4 * (size_t)n + i]; q.dz(i) = soa[5 * (size_t)n + i];
        q.tm(i) = soa[6 * (size_t)n + i];
    }
}
template <bool X>
__global__ __launch_bounds__(kBlock) void k_store_hits(Queue q, uint32_t n, SceneView sv, const double *hit_t,
                                                       const int32_t *hit_id, pt_hit *out) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        pt_hit h;
        const int32_t id = hit_id[i];
        if (id < 0) {
            h.t = 0.0; h.u = 0.0; h.v = 0.0;
            h.p[0] = h.p[1] = h.p[2] = 0.0;
            h.normal[0] = h.normal[1] = h.normal[2] = 0.0;
            h.prim = -1;
            h.material = -1;
        } else {
            const DevLeaf lf = sv.leaves[id & 0x0fffffff];
            const HitRec rec = finalize_hit<X, true>(sv, lf, d3(q.ox(i), q.oy(i), q.oz(i)),
                                                     d3(q.dx(i), q.dy(i), q.dz(i)), q.tm(i), hit_t[i], (id >> 28) & 7);
            h.t = rec.t; h.u = rec.u; h.v = rec.v;
            h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
            h.normal[0] = rec.n.x; h.normal[1] = rec.n.y; h.normal[2] = rec.n.z;
            h.prim = rec.prim;
            h.material = rec.mat;
        }
        out[i] = h;
    }
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
    bool need_uv = false;
    void *d_ops = nullptr, *d_xforms = nullptr, *d_leaves = nullptr, *d_lights = nullptr, *d_materials = nullptr,
         *d_textures = nullptr, *d_texels = nullptr, *d_perlin = nullptr;
    CameraDev cam{};

    // path state
    void *d_pool = nullptr; // one allocation carved into the arrays below
    Queue q[2]{};
    double *d_l[3] = {nullptr, nullptr, nullptr};
    double *d_hit_t = nullptr;
    int32_t *d_hit_id = nullptr;
    uint32_t *d_counts = nullptr;
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
    std::vector<uint32_t> h_counts, h_shard_counts;

    // launch shaping
    bool exact = false;            // PT_CFG_EXACT
    bool use_scan = false;         // wave-uniform extend for short op streams
    int scan_max_ops = 96;
    bool scan_list = false;        // ... over the stream with its BOX ops removed
    bool fuse = true;              // trace inside the shade kernel when scan_list applies (PTGPU_FUSE=0 disables)
    int list_max_leaves = 24;
    void *d_ops_list = nullptr;
    int32_t n_list_ops = 0;
    bool fuse_ok = false;          // the scene has a fused-kernel op list (d_ops_fused)
    uint32_t render_flags = 0;     // pt_render_args.flags of the render call in progress
    uint32_t persist_factor = 1;
    bool tree_pool = true;        // offline passes of tree scenes as one k_tree_pool launch
    uint32_t tree_tuning[4] = {16, 16, 32, 4}; // refill at / inner min / shade min / traversing min
    void *d_pool_consts = nullptr;
    std::deque<PoolConsts> h_pool_consts; // host copies of the passes in flight (cleared at every stream synchronisation of a render)
    bool exact_pool = true;       // exact mode uses the path pool too (short streams)
    bool path_pool = true;        // offline fused passes as one k_path_pool launch
    uint32_t pool_refill = 4;     // free path slots at which a wave generates new camera samples
    uint32_t pool_grid = 1;       // workgroups per resident slot
    uint32_t pool_chunk = 512;    // path ids a wave takes from the pass's counter at a time
    bool accel_pool = true;      // SAH traversal with lane refill (k_extend_accel_pool) instead of a ray per thread
    uint32_t accel_refill = 24;  // free lanes at which a wave takes new rays
    uint32_t accel_pool_from = 0; // first bounce traced by the pool form
    bool accel_pool_from_set = false;
    uint32_t accel_grid = 2;     // workgroups per resident slot
    uint32_t accel_inner = 24;   // lanes on inner nodes below which a wave turns to its leaves
    bool fuse_first = true;        // bounce 0 of an offline fused pass generates its own rays (PTGPU_FUSE_FIRST=0 disables)
    uint32_t persistent_max_paths = 1u << 23;
    bool persistent = true;        // fused scenes: one launch per pass (PTGPU_PERSISTENT=0: one launch per bounce)
    bool use_accel = true;         // fast mode: SAH tree for scenes that are not a short list (PTGPU_ACCEL=0 disables)
    bool accel_ok = false;
    void *d_accel_nodes = nullptr, *d_accel_prims = nullptr, *d_accel_parent = nullptr, *d_accel_levels = nullptr;
    std::vector<std::pair<int32_t, int32_t>> accel_levels; // refit order (deepest level first): index ranges of d_accel_levels, or node id ranges when that is null
    int accel_depth = 0;
    int32_t n_accel_prims = 0;
    void *d_ops_fused = nullptr;
    int32_t n_fused_ops = 0;
    std::vector<uint32_t> profile; // queue sizes per bounce of the last pass that was read back
    uint32_t profile_n = 0;        // its path count
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
    HIP_TRY(ctx, hipMalloc(dst, count * sizeof(T)));
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
    uint32_t shards = 8, shift = 3; // measured: 64 shards cost 18 % (scattered streams), 8 cost 2 % and leave 8x atomic headroom
    if (const char *e = std::getenv("PTGPU_SHARDS")) { // 1, 2, 4, ... 64
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 1 && v <= (long)kShards && (v & (v - 1)) == 0) {
            shards = (uint32_t)v;
            shift = 0;
            while ((1u << shift) < shards) ++shift;
        }
    }
    const uint32_t per_shard = ((ctx->capacity + kBlock * shards - 1) / (kBlock * shards)) * kBlock;
    ctx->capacity = per_shard * shards;
    ctx->shape.shards = shards;
    ctx->shape.shift = shift;
    ctx->shape.shard_cap = per_shard;
    const size_t n = ctx->capacity;
    // 2 queues x (10 f64 + key + 2 u32) + 3 f64 radiance + hit (f64 + i32)
    const size_t per_path = 2 * 96 + 3 * 8 + 8 + 4;
    const size_t n_arrays = 2 + 3 + 2;
    const size_t bytes = n * per_path + n_arrays * 256; // every array starts on a 256-byte boundary
    HIP_TRY(ctx, hipMalloc(&ctx->d_pool, bytes));
    char *p = (char *)ctx->d_pool;
    auto take = [&](size_t sz) {
        char *r = p;
        p += (sz + 255) & ~(size_t)255;
        return (void *)r;
    };
    for (int k = 0; k < 2; ++k) {
        Queue &q = ctx->q[k];
        q.base = (char *)take((n / 64) * (size_t)Queue::kTileBytes); // n is a multiple of kBlock
    }
    for (int k = 0; k < 3; ++k) ctx->d_l[k] = (double *)take(n * 8);
    ctx->d_hit_t = (double *)take(n * 8);
    ctx->d_hit_id = (int32_t *)take(n * 4);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_counts, (size_t)kMaxDepthSlots * kShards * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_stats, 2 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_trav, sizeof(TraversalCounters)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, 2 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_trav, 0, sizeof(TraversalCounters), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int ensure_image(pt_ctx *ctx, uint32_t w, uint32_t h) {
    if (ctx->image_external) {
        if (w != ctx->width || h != ctx->height)
            return set_error(ctx, PT_ERR_INVALID_ARG, "render size differs from the bound image");
    } else if (!ctx->d_image_owned || w != ctx->width || h != ctx->height) {
        if (ctx->d_image_owned) (void)hipFree(ctx->d_image_owned);
        ctx->d_image_owned = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_image_owned, (size_t)w * h * 4 * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_image_owned, 0, (size_t)w * h * 4 * sizeof(float), ctx->stream));
        ctx->d_image = ctx->d_image_owned;
        ctx->width = w;
        ctx->height = h;
        if (ctx->d_frames) (void)hipFree(ctx->d_frames);
        ctx->d_frames = nullptr;
    }
    if (!ctx->d_frames) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_frames, (size_t)w * h * sizeof(uint32_t)));
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
    for (int k = 0; k < 3; ++k) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_acc[k], pixels * sizeof(double)));
    ctx->acc_capacity = pixels;
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
    if (!X && ctx->accel_ok && a.counters == nullptr && ctx->accel_pool && !(ctx->accel_pool_from > a.depth && a.key_mode == 0)) {
        grid = std::min(grid, ctx->n_cu * (MEDIA ? 4 : 5) * (int)ctx->accel_grid); // what is resident at once
        hipLaunchKernelGGL((k_extend_accel_pool<MEDIA>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                           (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a, ctx->accel_refill | (ctx->accel_inner << 8));
    } else if (!X && ctx->accel_ok && a.counters == nullptr) {
        hipLaunchKernelGGL((k_extend_accel<MEDIA>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view,
                           (const AccelNode *)ctx->d_accel_nodes, (const DevOp *)ctx->d_accel_prims, a);
    } else if (!X && !FRAMES && ctx->use_scan && ctx->scan_list && ctx->fuse_ok && ctx->fuse && a.counters == nullptr) {
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
        hipLaunchKernelGGL((k_extend<X, FRAMES, MEDIA>), dim3(grid), dim3(kBlock), 0, ctx->stream, q, ctx->view, a);
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
template <bool X, bool UV, int FUSE>
void launch_shade_xuf(pt_ctx *ctx, int grid, const Queue &qi, const Queue &qo, uint32_t d, uint32_t max_depth) {
    SceneView v = ctx->view;
    if (FUSE) {
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
    }
    hipLaunchKernelGGL((k_shade<X, UV, FUSE>), dim3(grid), dim3(kBlock), 0, ctx->stream, qi, qo, ctx->shape,
                       ctx->d_counts + (size_t)d * kShards, ctx->d_counts + (size_t)(d + 1) * kShards, v, ctx->d_hit_t,
                       ctx->d_hit_id, ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], d, max_depth, ctx->render_flags);
}
// fused = trace inside the shade kernel (fast mode, box-free list, no transformed subtrees)
void launch_shade(pt_ctx *ctx, int grid, const Queue &qi, const Queue &qo, uint32_t d, uint32_t max_depth, bool fused) {
    if (ctx->exact) {
        if (ctx->need_uv) launch_shade_xuf<true, true, 0>(ctx, grid, qi, qo, d, max_depth);
        else launch_shade_xuf<true, false, 0>(ctx, grid, qi, qo, d, max_depth);
    } else if (!fused) {
        if (ctx->need_uv) launch_shade_xuf<false, true, 0>(ctx, grid, qi, qo, d, max_depth);
        else launch_shade_xuf<false, false, 0>(ctx, grid, qi, qo, d, max_depth);
    } else if (ctx->view.has_medium) {
        if (ctx->need_uv) launch_shade_xuf<false, true, 2>(ctx, grid, qi, qo, d, max_depth);
        else launch_shade_xuf<false, false, 2>(ctx, grid, qi, qo, d, max_depth);
    } else {
        if (ctx->need_uv) launch_shade_xuf<false, true, 1>(ctx, grid, qi, qo, d, max_depth);
        else launch_shade_xuf<false, false, 1>(ctx, grid, qi, qo, d, max_depth);
    }
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
    const bool fused = ctx->fuse && ctx->scan_list && ctx->fuse_ok && !ctx->exact && !count;
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

    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counts, 0, (size_t)(iters + 1) * kShards * sizeof(uint32_t), st));
    const int g_all = grid_for(ctx, n);
    // Measured on the Cornell box, one launch per pass against one per bounce: 2^20 paths +53 %, 2^22 +13 %,
    // 2^24 -3 %, 2^26 -7 % (the global compaction keeps every wave's tiles full and spreads the tail of long
    // paths over the whole chip); a 1080p realtime frame (0.4 M paths) +11 %.
    // (realtime frames, 0.4 M paths after the pixel skip, measured 16 % slower through the pool than through the
    // persistent form below: four of five generated samples are skipped)
    const bool pool = fused && ctx->path_pool && !realtime;
    // exact mode, short streams: the same pool around the wave-uniform walk of the reference's own tree
    const bool xpool = ctx->exact && ctx->exact_pool && ctx->path_pool && ctx->use_scan && !count && !realtime &&
                       ctx->view.n_leaves <= kLdsLeaves;
    const bool tree = !fused && !realtime && ctx->tree_pool && ctx->accel_ok && !ctx->exact && !count;
    const bool persistent = fused && !pool && ctx->persistent && n <= ctx->persistent_max_paths;
    // the persistent kernel deals contiguous runs of tiles to its waves: one shard, entry = path id
    const QueueShape gen_shape = persistent ? QueueShape{1u, 0u, ctx->capacity} : ctx->shape;
    // offline passes of fused scenes with a launch per bounce: bounce 0 generates its own rays
    const bool gen0 = fused && !pool && !persistent && !realtime && ctx->fuse_first;
    HIP_TRY(ctx, mark());
    auto send_pool_consts = [&]() -> hipError_t {
        if (!ctx->d_pool_consts)
            if (hipError_t e = hipMalloc(&ctx->d_pool_consts, sizeof(PoolConsts)); e != hipSuccess) return e;
        ctx->h_pool_consts.emplace_back(PoolConsts{ctx->cam, pp}); // must outlive the asynchronous copy
        return hipMemcpyAsync(ctx->d_pool_consts, &ctx->h_pool_consts.back(), sizeof(PoolConsts), hipMemcpyHostToDevice, st);
    };
    if (pool || tree || xpool) {
        // no ray generation launch: the pool kernels make their own camera samples
    } else if (gen0) {
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
        const bool media = ctx->view.has_medium != 0;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(g_all), dim3(kBlock), 0, st, ctx->q[1], ctx->shape, ctx->d_counts, v, ctx->cam,
                               pp, ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->render_flags);
        };
        if (media && ctx->need_uv) launch(k_first_bounce<true, 2>);
        else if (media) launch(k_first_bounce<false, 2>);
        else if (ctx->need_uv) launch(k_first_bounce<true, 1>);
        else launch(k_first_bounce<false, 1>);
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
        if (media && ctx->need_uv) launch(k_trace_persistent<true, 2>);
        else if (media) launch(k_trace_persistent<false, 2>);
        else if (ctx->need_uv) launch(k_trace_persistent<true, 1>);
        else launch(k_trace_persistent<false, 1>);
        HIP_TRY(ctx, mark());
    }
    if (pool || xpool) {
        const uint32_t tiles = (n + 63u) / 64u;
        const uint32_t resident = (uint32_t)ctx->n_cu * (xpool ? PT_EXACT_POOL_WAVES : PT_SHADE_WAVES) * ctx->pool_grid;
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(resident, (tiles + kWavesPerBlock - 1) / kWavesPerBlock));
        // a small pass is cut into at least four chunks per wave so that the waves finish together
        const uint32_t fair = (n / (grid * kWavesPerBlock * 4u) + 63u) & ~63u;
        const uint32_t chunk = std::max(64u, std::min(ctx->pool_chunk, fair));
        SceneView v = ctx->view;
        v.ops = (const DevOp *)ctx->d_ops_fused;
        v.n_traversal_ops = ctx->n_fused_ops;
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        const bool media = ctx->view.has_medium != 0;
        HIP_TRY(ctx, send_pool_consts());
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, v, (const PoolConsts *)ctx->d_pool_consts, n,
                               pp.max_depth, ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags,
                               ctx->pool_refill, chunk);
        };
        if (xpool) {
            v = ctx->view; // the reference's own tree
            if (media && ctx->need_uv) launch(k_path_pool<true, 2, true>);
            else if (media) launch(k_path_pool<false, 2, true>);
            else if (ctx->need_uv) launch(k_path_pool<true, 1, true>);
            else launch(k_path_pool<false, 1, true>);
        } else if (media && ctx->need_uv) launch(k_path_pool<true, 2>);
        else if (media) launch(k_path_pool<false, 2>);
        else if (ctx->need_uv) launch(k_path_pool<true, 1>);
        else launch(k_path_pool<false, 1>);
        HIP_TRY(ctx, mark());
    }
    if (tree) {
        const uint32_t tiles = (n + 63u) / 64u;
        const uint32_t resident = (uint32_t)ctx->n_cu * (ctx->need_uv ? PT_TREE_UV_WAVES : PT_SHADE_WAVES) * ctx->pool_grid;
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(resident, (tiles + kWavesPerBlock - 1) / kWavesPerBlock));
        const uint32_t fair = (n / (grid * kWavesPerBlock * 4u) + 63u) & ~63u;
        const uint32_t chunk = std::max(64u, std::min(ctx->pool_chunk, fair));
        HIP_TRY(ctx, send_pool_consts());
        const uint32_t tuning = ctx->tree_tuning[0] | (ctx->tree_tuning[1] << 8) | (ctx->tree_tuning[2] << 16) | (ctx->tree_tuning[3] << 24);
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        const bool media = ctx->view.has_medium != 0;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, ctx->view, (const AccelNode *)ctx->d_accel_nodes,
                               (const DevOp *)ctx->d_accel_prims, (const PoolConsts *)ctx->d_pool_consts, n, pp.max_depth,
                               ctx->d_l[0], ctx->d_l[1], ctx->d_l[2], ctx->d_counts, ctx->render_flags, tuning, chunk);
        };
        if (media && ctx->need_uv) launch(k_tree_pool<true, true>);
        else if (media) launch(k_tree_pool<false, true>);
        else if (ctx->need_uv) launch(k_tree_pool<true, false>);
        else launch(k_tree_pool<false, false>);
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
        if (!fused) launch_extend(ctx, g_d, qi, ea);
        HIP_TRY(ctx, mark());
        HIP_TRY(ctx, mark());
        if (!(gen0 && d == 0)) launch_shade(ctx, g_d, qi, qo, d, pp.max_depth, fused);
        HIP_TRY(ctx, mark());
    }
    HIP_TRY(ctx, mark());
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, pp.cpix)), dim3(kBlock), 0, st, ctx->d_l[0], ctx->d_l[1],
                       ctx->d_l[2], ctx->d_acc[0], ctx->d_acc[1], ctx->d_acc[2], ctx->d_image, ctx->d_frames, pp,
                       ctx->d_counts, ctx->d_stats);
    HIP_TRY(ctx, mark());
    HIP_TRY(ctx, hipGetLastError());
    ctx->stats.paths += n;
    ctx->stats.launches_raygen += (gen0 || pool || tree || xpool) ? 0 : 1;
    ctx->stats.launches_extend += (fused || tree || xpool) ? 0 : iters;
    ctx->stats.launches_shade += one_launch ? 1 : (gen0 ? iters - 1 : iters);
    ctx->stats.launches_resolve += 1;
    ctx->stats.last_pass_form = tree ? PT_PASS_TREE_POOL : (pool || xpool) ? PT_PASS_PATH_POOL : persistent ? PT_PASS_PERSISTENT
                                : fused ? PT_PASS_PER_BOUNCE_FUSED : PT_PASS_PER_BOUNCE;
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
        ctx->stats.ms_raygen += elapsed(k); // with gen0 this launch is k_first_bounce: ray generation AND bounce 0
        k += 2;
        for (uint32_t d = 0; d < (one_launch ? 1u : iters); ++d) {
            ctx->stats.ms_extend += elapsed(k);
            k += 2;
            ctx->stats.ms_shade += elapsed(k);
            k += 2;
        }
        // segments of the launches timed as extend / shade (bounce 0 belongs to k_first_bounce when it ran)
        for (uint32_t d = gen0 ? 1u : 0u; d < iters; ++d) ctx->stats.profiled_segments += ctx->h_counts[d];
        ctx->stats.ms_resolve += elapsed(k);
        ctx->stats.profiled_paths += n;
        ctx->stats.profiled_passes += 1;
        ctx->stats.profiled_bounce_launches += one_launch ? 1 : (gen0 ? iters - 1 : iters);
    }
    return PT_OK;
}

} // namespace

// ================================================================================================================
// extern "C"
// ================================================================================================================
extern "C" {

int pt_create(const pt_config *cfg, pt_ctx **out) {
    if (!cfg || !out || cfg->struct_size != sizeof(pt_config)) return PT_ERR_INVALID_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return PT_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= n_dev) return PT_ERR_INVALID_ARG;
    pt_ctx *ctx = new (std::nothrow) pt_ctx();
    if (!ctx) return PT_ERR_OUT_OF_MEMORY;
    ctx->device = cfg->device;
    ctx->seed = cfg->seed;
    ctx->flags = cfg->flags;
    ctx->exact = (cfg->flags & PT_CFG_EXACT) != 0;
    if (const char *e = std::getenv("PTGPU_SCAN_MAX_OPS")) ctx->scan_max_ops = (int)std::strtol(e, nullptr, 10);
    if (const char *e = std::getenv("PTGPU_LIST_MAX_LEAVES")) ctx->list_max_leaves = (int)std::strtol(e, nullptr, 10);
    if (const char *e = std::getenv("PTGPU_FUSE")) ctx->fuse = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_ACCEL")) ctx->use_accel = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_FUSE_FIRST")) ctx->fuse_first = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_PERSISTENT")) ctx->persistent = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_PERSISTENT_MAX_PATHS")) ctx->persistent_max_paths = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char *e = std::getenv("PTGPU_ACCEL_POOL")) ctx->accel_pool = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_ACCEL_POOL_FROM")) {
        ctx->accel_pool_from = (uint32_t)std::max(0l, std::strtol(e, nullptr, 10));
        ctx->accel_pool_from_set = true;
    }
    if (const char *e = std::getenv("PTGPU_ACCEL_GRID")) ctx->accel_grid = (uint32_t)std::min(8l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_ACCEL_INNER")) ctx->accel_inner = (uint32_t)std::min(65l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_ACCEL_REFILL")) ctx->accel_refill = (uint32_t)std::min(64l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_TREE_POOL")) ctx->tree_pool = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_TREE_TUNING")) { // "refill,inner,shade,trav"
        unsigned v[4];
        if (std::sscanf(e, "%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3]) == 4)
            for (int k = 0; k < 4; ++k) ctx->tree_tuning[k] = std::min(255u, std::max(k == 3 ? 0u : 1u, v[k]));
    }
    if (const char *e = std::getenv("PTGPU_EXACT_POOL")) ctx->exact_pool = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_PATH_POOL")) ctx->path_pool = std::strtol(e, nullptr, 10) != 0;
    if (const char *e = std::getenv("PTGPU_POOL_REFILL")) ctx->pool_refill = (uint32_t)std::min(64l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_POOL_CHUNK")) ctx->pool_chunk = (uint32_t)std::min(1l << 20, std::max(64l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_POOL_GRID")) ctx->pool_grid = (uint32_t)std::min(256l, std::max(1l, std::strtol(e, nullptr, 10)));
    if (const char *e = std::getenv("PTGPU_PERSIST_FACTOR")) ctx->persist_factor = (uint32_t)std::max(1l, std::strtol(e, nullptr, 10));
    ctx->capacity = cfg->max_paths_in_flight ? cfg->max_paths_in_flight : kDefaultPathsInFlight;
    if (ctx->capacity < 1024) ctx->capacity = 1024;
    if (const char *e = std::getenv("PTGPU_PROFILE_PERIOD")) {
        long v = std::strtol(e, nullptr, 10);
        if (v >= 1) ctx->profile_period = (uint32_t)v;
    }
    auto bail = [&](int code) {
        pt_destroy(ctx);
        return code;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) return bail(PT_ERR_HIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(PT_ERR_HIP);
    int rc = alloc_path_state(ctx);
    if (rc != PT_OK) return bail(rc);
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *ptrs[] = {ctx->d_ops,     ctx->d_xforms,  ctx->d_leaves, ctx->d_lights, ctx->d_materials, ctx->d_textures,
                    ctx->d_texels,  ctx->d_perlin,  ctx->d_pool,   ctx->d_counts, ctx->d_stats,     ctx->d_trav,
                    ctx->d_ops_list, ctx->d_ops_fused, ctx->d_accel_nodes, ctx->d_accel_prims, ctx->d_accel_parent, ctx->d_accel_levels,
                    ctx->d_image_owned, ctx->d_frames, ctx->d_acc[0], ctx->d_acc[1], ctx->d_acc[2], ctx->d_pool_consts};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

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
    v.has_isotropic = 0;
    for (int i = 0; i < scene->n_materials; ++i)
        if (scene->materials[i].kind == PT_M_ISOTROPIC) v.has_isotropic = 1;
    ctx->need_uv = false;
    for (int i = 0; i < scene->n_textures; ++i)
        if (scene->textures[i].kind == PT_T_IMAGE) ctx->need_uv = true;
    ctx->use_scan = f.n_traversal_ops <= ctx->scan_max_ops;
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
    if (!ctx->exact && !ctx->scan_list && ctx->use_accel) {
        Accel acc;
        std::string why;
        if (build_accel(f, acc, why)) {
            if ((rc = upload(ctx, &ctx->d_accel_nodes, acc.nodes.data(), acc.nodes.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_prims, acc.prims.data(), acc.prims.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_parent, acc.parent_slot.data(), acc.parent_slot.size())) != PT_OK) return rc;
            if ((rc = upload(ctx, &ctx->d_accel_levels, acc.by_level.data(), acc.by_level.size())) != PT_OK) return rc;
            ctx->accel_levels.clear();
            for (size_t l = 0; l + 1 < acc.level_start.size(); ++l)
                ctx->accel_levels.emplace_back(acc.level_start[l], acc.level_start[l + 1]);
            ctx->accel_depth = acc.depth;
            ctx->n_accel_prims = (int32_t)acc.prims.size();
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
    if (ctx->d_image_owned) (void)hipFree(ctx->d_image_owned);
    ctx->d_image_owned = nullptr;
    ctx->image_external = true;
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
    ctx->render_flags = a->flags;

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
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->h_pool_consts.clear();
    HIP_TRY(ctx, hipGetLastError());
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
    HIP_TRY(ctx, hipMemcpyAsync(rgba, ctx->d_image, n_floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

static int intersect_common(pt_ctx *ctx, size_t n, double t_min, double t_max, uint64_t medium_seed, int repeats,
                            double *mean_ms_out) {
    const bool count = (ctx->flags & PT_CFG_COUNT_TRAVERSAL) != 0;
    const uint32_t n32 = (uint32_t)n;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_counts, &n32, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    const int g = grid_for(ctx, n);
    ExtendArgs ea;
    ea.counts = ctx->d_counts;
    ea.sh = QueueShape{1u, 0u, ctx->capacity}; // one shard: slot == ray index
    ea.t_min = t_min;
    ea.t_max = t_max;
    ea.medium_seed = medium_seed;
    ea.key_mode = 1;
    ea.depth = 0;
    ea.hit_t = ctx->d_hit_t;
    ea.hit_id = ctx->d_hit_id;
    ea.counters = count ? ctx->d_trav : nullptr;
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < repeats; ++r) launch_extend(ctx, g, ctx->q[0], ea);
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (mean_ms_out) *mean_ms_out = (double)ms / (double)repeats;
    ctx->stats.segments += 0;
    return PT_OK;
}

int pt_intersect(pt_ctx *ctx, const pt_ray *rays, size_t n, double t_min, double t_max, uint64_t medium_seed,
                 pt_hit *out) {
    if (!ctx || (n && (!rays || !out))) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_intersect before pt_scene_upload");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    // batches larger than the queue capacity are processed in slices; the medium key uses the slice-local index
    // offset by the slice start so results do not depend on the slicing
    if (n > ctx->capacity) return set_error(ctx, PT_ERR_TOO_LARGE, "ray batch larger than the paths-in-flight budget");
    void *d_rays = nullptr, *d_hits = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_rays, n * sizeof(pt_ray)));
    if (hipMalloc(&d_hits, n * sizeof(pt_hit)) != hipSuccess) {
        (void)hipFree(d_rays);
        return set_error(ctx, PT_ERR_OUT_OF_MEMORY, "hipMalloc hits");
    }
    int rc = PT_OK;
    do {
        if (hipMemcpyAsync(d_rays, rays, n * sizeof(pt_ray), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = set_error(ctx, PT_ERR_HIP, "H2D rays");
            break;
        }
        const int g = grid_for(ctx, n);
        hipLaunchKernelGGL(k_load_rays, dim3(g), dim3(kBlock), 0, ctx->stream, (const pt_ray *)d_rays, (uint32_t)n,
                           ctx->q[0]);
        rc = intersect_common(ctx, n, t_min, t_max, medium_seed, 1, nullptr);
        if (rc != PT_OK) break;
        if (ctx->exact)
            hipLaunchKernelGGL(k_store_hits<true>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->q[0], (uint32_t)n,
                               ctx->view, ctx->d_hit_t, ctx->d_hit_id, (pt_hit *)d_hits);
        else
            hipLaunchKernelGGL(k_store_hits<false>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->q[0], (uint32_t)n,
                               ctx->view, ctx->d_hit_t, ctx->d_hit_id, (pt_hit *)d_hits);
        if (hipMemcpyAsync(out, d_hits, n * sizeof(pt_hit), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = set_error(ctx, PT_ERR_HIP, "D2H hits");
            break;
        }
    } while (0);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    return rc;
}

int pt_intersect_device(pt_ctx *ctx, const double *d_rays_soa, size_t n, double t_min, double t_max,
                        uint64_t medium_seed, int repeats, double *mean_ms_out) {
    if (!ctx || !d_rays_soa || n == 0 || repeats <= 0) return PT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return set_error(ctx, PT_ERR_NO_SCENE, "pt_intersect_device before pt_scene_upload");
    if (n > ctx->capacity) return set_error(ctx, PT_ERR_TOO_LARGE, "ray batch larger than the paths-in-flight budget");
    (void)hipSetDevice(ctx->device);
    hipLaunchKernelGGL(k_load_rays_soa, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, d_rays_soa, (uint32_t)n,
                       ctx->q[0]);
    return intersect_common(ctx, n, t_min, t_max, medium_seed, repeats, mean_ms_out);
}

int pt_tonemap_device(pt_ctx *ctx, uint8_t *rgb8_out, size_t n_bytes) {
    if (!ctx || !rgb8_out) return PT_ERR_INVALID_ARG;
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    const size_t need = (size_t)ctx->width * ctx->height * 3;
    if (n_bytes < need) return set_error(ctx, PT_ERR_INVALID_ARG, "rgb8_out too small");
    (void)hipSetDevice(ctx->device);
    uint8_t *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_out, need));
    const uint32_t n = ctx->width * ctx->height;
    hipLaunchKernelGGL(k_tonemap_rgb8, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->d_image, ctx->width,
                       ctx->height, d_out);
    hipError_t e = hipMemcpyAsync(rgb8_out, d_out, need, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return set_error(ctx, PT_ERR_HIP, hipGetErrorString(e));
    return PT_OK;
}

int pt_display_transform(pt_ctx *ctx, float exposure, float *rgba_out, size_t n_floats) {
    if (!ctx || !rgba_out) return PT_ERR_INVALID_ARG;
    if (!ctx->d_image || ctx->width == 0) return set_error(ctx, PT_ERR_INVALID_ARG, "no image: render first");
    const size_t need = (size_t)ctx->width * ctx->height * 4;
    if (n_floats < need) return set_error(ctx, PT_ERR_INVALID_ARG, "rgba_out too small");
    (void)hipSetDevice(ctx->device);
    float *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_out, need * sizeof(float)));
    const uint32_t n = ctx->width * ctx->height;
    hipLaunchKernelGGL(k_display_transform, dim3(grid_for(ctx, n)), dim3(kBlock), 0, ctx->stream, ctx->d_image,
                       ctx->width, ctx->height, exposure, d_out);
    hipError_t e = hipMemcpyAsync(rgba_out, d_out, need * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return set_error(ctx, PT_ERR_HIP, hipGetErrorString(e));
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
    return PT_OK;
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
        void *prims = nullptr, *nodes = nullptr, *parent = nullptr;
        ~Held() {
            for (void *p : {keys[0], keys[1], vals[0], vals[1], temp, ranges, counter, prims, nodes, parent})
                if (p) (void)hipFree(p);
        }
    } h;
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(ctx, hipMalloc(&h.keys[k], (size_t)n * sizeof(uint64_t)));
        HIP_TRY(ctx, hipMalloc(&h.vals[k], (size_t)n * sizeof(uint32_t)));
    }
    HIP_TRY(ctx, hipMalloc(&h.prims, (size_t)n * sizeof(DevOp)));
    // at most n - 1 nodes (every leaf a single primitive)
    HIP_TRY(ctx, hipMalloc(&h.nodes, (size_t)n * sizeof(AccelNode)));
    HIP_TRY(ctx, hipMalloc(&h.parent, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc(&h.ranges, (size_t)n * sizeof(BuildRange)));
    HIP_TRY(ctx, hipMalloc(&h.counter, sizeof(int32_t)));
    hipLaunchKernelGGL(k_accel_prim_codes, dim3(grid_for(ctx, (uint32_t)n)), dim3(kBlock), 0, ctx->stream,
                       (const DevOp *)ctx->d_accel_prims, ctx->view, n, time0, time1, D3{lo[0], lo[1], lo[2]},
                       D3{scale[0], scale[1], scale[2]}, (uint64_t *)h.keys[0], (uint32_t *)h.vals[0]);
    size_t temp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint64_t *)h.keys[0], (uint64_t *)h.keys[1],
                                                    (const uint32_t *)h.vals[0], (uint32_t *)h.vals[1], n, 0, 63, ctx->stream));
    HIP_TRY(ctx, hipMalloc(&h.temp, std::max<size_t>(temp_bytes, 16)));
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(h.temp, temp_bytes, (const uint64_t *)h.keys[0], (uint64_t *)h.keys[1],
                                                    (const uint32_t *)h.vals[0], (uint32_t *)h.vals[1], n, 0, 63, ctx->stream));
    hipLaunchKernelGGL(k_accel_gather_prims, dim3(grid_for(ctx, (uint64_t)n * 4)), dim3(kBlock), 0, ctx->stream,
                       (const DevOp *)ctx->d_accel_prims, (const uint32_t *)h.vals[1], (DevOp *)h.prims, n);
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
    std::swap(ctx->d_accel_nodes, h.nodes);
    std::swap(ctx->d_accel_parent, h.parent);
    if (ctx->d_accel_levels) (void)hipFree(ctx->d_accel_levels);
    ctx->d_accel_levels = nullptr; // node ids are level-ordered already
    ctx->accel_levels.assign(levels.rbegin(), levels.rend());
    ctx->accel_depth = (int)levels.size();
    ctx->accel_ok = false; // until the boxes are in
    if ((rc = refit_levels(ctx, time0, time1)) != PT_OK) return rc;
    ctx->accel_ok = true;
    return PT_OK;
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

} // extern "C"
