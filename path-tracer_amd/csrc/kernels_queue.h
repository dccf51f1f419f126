// Launch constants, the path-state queue layout, wave64 compaction and ray generation.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

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
