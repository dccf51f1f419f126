// The path pool kernels: one launch per offline pass, a path in one lane's registers from camera sample to last vertex.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

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
    AccelGrid grid; // of the 32-byte nodes, for the kernels that read those
    // Where a ray starts in the 32-byte nodes: on `start_node`, with `start_push` on its stack when start_sp is 1. A small
    // tree whose root holds a leaf (the scene-wide primitives) and the tree over the rest has its grid over the rest
    // alone; the root's own boxes, which that grid cannot hold, are not tested: both children count as hit.
    int32_t start_node, start_push; // (start_node < 0, a leaf: one stack entry)
    // A realtime frame of one sample per pixel traced in one pass (resolve_inline != 0): every path of the frame owns its
    // pixel, so the path that ends blends it into the image itself (blend_pixel) and no resolve launch follows.
    float *image;
    uint32_t *frames_per_pixel;
    unsigned long long *stats; // [0] segments, [1] NaN pixels
    uint32_t resolve_inline, reserved;
};
// Puts a pass's constants where the pool kernels read them. They arrive as kernel arguments, i.e. with the launch packet:
// a copy from pageable host memory costs the host a staging copy and a wait of its own per pass, which a realtime frame
// of a fifth of a millisecond notices.
// The same launch clears the pass's counters (queue sizes, the chunk counter of the pool kernels).
__global__ void k_set_pool_consts(PoolConsts value, PoolConsts *dst, uint32_t *counts, uint32_t n_counts) {
    constexpr uint32_t kWords = sizeof(PoolConsts) / sizeof(uint32_t);
    static_assert(sizeof(PoolConsts) % sizeof(uint32_t) == 0, "copied word by word");
    const uint32_t *src = (const uint32_t *)&value;
    for (uint32_t k = threadIdx.x; k < kWords; k += blockDim.x) ((uint32_t *)dst)[k] = src[k];
    for (uint32_t k = threadIdx.x; k < n_counts; k += blockDim.x) counts[k] = 0u;
}
// Where a pool kernel reads its pass constants. An offline pass reads the copy in device memory that k_set_pool_consts
// wrote. A realtime frame is a fifth of a millisecond, of which every operation on the stream costs some fifteen
// microseconds: its constants travel with the launch packet itself, as the kernel's FIRST argument, and are read from the
// kernarg segment where they are needed (the parameter is never named: named, its sixty words would sit in scalar
// registers for the whole kernel), so the frame needs no launch of its own to deliver them.
template <bool RT> __device__ __forceinline__ const PoolConsts *pool_consts_of(const PoolConsts *in_memory) {
    if constexpr (RT) return (const PoolConsts *)__builtin_amdgcn_kernarg_segment_ptr();
    else return in_memory;
}
// What k_resolve does for one pixel of a realtime frame of one sample (src/trace.rs:166-177), by the lane whose path ended:
// the same operations on the same values (the pass's first and only sum starts from 0.0; a division by one sample is
// exact), so the image is the one the resolve kernel writes. Returns 1 if the pixel is NaN afterwards.
__device__ __forceinline__ uint32_t blend_pixel(const PoolConsts *consts, uint32_t pixel, D3 sum) {
    const PoolConsts *pc = consts;
    asm volatile("" : "+s"(pc)); // keeps the loads below in this block
    float *px = load_uniform(&pc->image) + (size_t)pixel * 4;
    uint32_t *frames = load_uniform(&pc->frames_per_pixel);
    const double cx = 0.0 + sum.x, cy = 0.0 + sum.y, cz = 0.0 + sum.z;
    uint32_t c = frames[pixel];
    c += (c <= 1000u) ? 1u : 0u; // src/trace.rs:149
    frames[pixel] = c;
    const float weight = 1.0f / (float)c; // src/trace.rs:171-177
    const float one_minus_weight = 1.0f - weight;
    const float r = (float)cx * weight + px[0] * one_minus_weight;
    const float g = (float)cy * weight + px[1] * one_minus_weight;
    const float b = (float)cz * weight + px[2] * one_minus_weight;
    px[0] = r;
    px[1] = g;
    px[2] = b;
    return (r != r || g != g || b != b) ? 1u : 0u;
}
// the pass's NaN pixels, added to the statistics once per wave
__device__ __forceinline__ void add_nan_pixels(const PoolConsts *consts, uint32_t nan_pixels) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nan_pixels += (uint32_t)__shfl_xor((int)nan_pixels, off);
    if ((threadIdx.x & 63u) == 0 && nan_pixels) atomicAdd(load_uniform(&consts->stats) + 1, (unsigned long long)nan_pixels);
}
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
#ifndef PT_MEDIA_POOL_WAVES
#define PT_MEDIA_POOL_WAVES 3 // the variant with media inside (FUSE == 2): at 4 waves per SIMD it holds 196 B of scratch per lane (smoke box: 1 570 -> 2 220 Mpaths/s)
#endif
#ifndef PT_EXACT_FLAT_POOL_WAVES
#define PT_EXACT_FLAT_POOL_WAVES 4 // exact mode, stream without ENTER / EXIT ops (FUSE 3, 4): no frame ray to carry; Cornell box 3 -> 2 298, 4 -> 2 447, 5 -> 2 140 Mpaths/s; with a medium 3 and 4 tie
#endif
constexpr int pool_waves(bool exact, int fuse) {
    return exact ? (fuse >= 3 ? PT_EXACT_FLAT_POOL_WAVES : PT_EXACT_POOL_WAVES) : (fuse == 2 ? PT_MEDIA_POOL_WAVES : PT_SHADE_WAVES);
}
// X: exact mode. The walk is the wave-uniform scan of the reference's own threaded tree (every box test, the
// reference's order), the arithmetic is M<true>'s; sv.ops is then the full stream, not the fused list.
// RT: a realtime frame (src/trace.rs:144-146: four of five pixels are skipped, by a draw keyed on seed, frame and pixel).
// The path ids of a frame cover every pixel; before a wave makes its next 64 camera samples it evaluates the skip draw
// of 64 ids at a time with all its lanes and collects the ids that survive, so that no sample is made for a pixel the
// frame skips.
template <int TEX, int FUSE, bool X = false, bool RT = false>
__global__ __launch_bounds__(kBlock, pool_waves(X, FUSE)) void k_path_pool(PoolConsts consts_by_value, SceneView sv, const PoolConsts *consts_in_memory, uint32_t n,
                                                                      uint32_t max_depth, double *lx, double *ly,
                                                                      double *lz, uint32_t *counts, uint32_t flags,
                                                                      uint32_t refill_arg, uint32_t chunk) {
    const PoolConsts *consts = pool_consts_of<RT>(consts_in_memory);
    // RT, one sample per pixel in one pass: a path carries its pixel's index instead of its path id and blends the pixel
    // itself when it ends (needs the chunks dealt by wave index: the counters are not touched at all then)
    bool inline_resolve = false;
    if constexpr (RT) inline_resolve = load_uniform(&consts->resolve_inline) != 0u && (refill_arg >> 31) != 0u;
    uint32_t nan_pixels = 0;
    __shared__ PoolLds S;
    __shared__ uint32_t live[RT ? kWavesPerBlock : 1][128]; // RT: surviving path ids of the wave, not yet made into samples
    uint32_t live_n = 0;
    // Scenes with media: the rejection loop of random_in_unit_sphere (every Isotropic scatter, every Metal reflection) is
    // run by the wave together (wave_unit_sphere); its table of wanting lanes. (Smoke box +4.2 % in reference semantics,
    // +1.7 % in book semantics, same images. The tree pool keeps the serial loop: its variant for the final scene is a
    // hundred registers over its budget already, and the sampler's live values cost it 40 B more scratch per lane, -1.5 %.)
    constexpr bool kWaveSphere = FUSE == 2 || FUSE == 4;
    __shared__ uint32_t sphere_src[kWaveSphere ? kWavesPerBlock : 1][64];
    const uint32_t refill_at = min(max(refill_arg & 0xffu, 1u), 64u); // outside [1, 64] the loops below could not end
    // RT: the launch has a chunk for every wave (bit 31 of refill_arg): wave k takes chunk k and no counter is involved. A
    // frame is short, and 4 096 waves queueing at one atomic address were a sixth of it.
    const bool static_chunks = (refill_arg >> 31) != 0u;
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
    bool exhausted = false, started = false;
    bool has = false;
    D3 o = d3(0.0, 0.0, 0.0), d = d3(0.0, 0.0, 1.0), beta = d3(0.0, 0.0, 0.0), sum = d3(0.0, 0.0, 0.0);
    double time = 0.0;
    uint64_t pkey = 0;
    uint32_t pid = 0, rn = 0, depth = 0, segments = 0;
#ifdef PT_POOL_TIMING
    long long tm_refill = 0, tm_walk = 0, tm_shade = 0, tm_rounds = 0, tm_lanes = 0, tm0 = clock64();
    for (int k = lane; k < 3 * kShadePhases + 1; k += 64) shade_clock()[k] = 0;
#endif
    for (;;) {
        bool drained = false;
        for (;;) { // hand camera samples to free lanes
            const unsigned long long idle = __ballot(!has);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle < refill_at) break;
            if (stash_n == 0) {
                auto next_chunk = [&]() { // the wave's next chunk of path ids, when the current one is used up
                    if (cur >= end && !exhausted) {
                        uint32_t first = 0;
                        if (RT && static_chunks) { // its one chunk, then none
                            first = started ? n : min((blockIdx.x * kWavesPerBlock + wv) * chunk, n);
                            started = true;
                        } else if (lane == 0) first = atomicAdd(&counts[(size_t)(max_depth + 1) * kShards], chunk);
                        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
                        exhausted = first >= n;
                        cur = min(first, n);
                        end = min(cur + chunk, n); // chunk is far below 2^32 - n
                    }
                };
                const PoolConsts *pc = consts;
                asm volatile("" : "+s"(pc)); // keeps the loads below in this block
                uint32_t m, id;
                if constexpr (RT) {
                    const PassParams ppr = load_uniform(&pc->pp);
                    while (live_n < 64u) { // skip draws of the next 64 ids; the survivors join the list (at most 64 + 64)
                        next_chunk();
                        if (cur >= end) break;
                        const uint32_t h = cur + lane;
                        bool alive = false;
                        if (h < end) {
                            const uint32_t p = h % ppr.cpix;
                            const uint64_t pixel = (uint64_t)(ppr.cy0 + (p / ppr.cw) * ppr.row_step) * ppr.width + (ppr.cx0 + p % ppr.cw);
                            alive = !(pixel_skip_draw(ppr.seed, ppr.frame, pixel) < 0.8); // src/trace.rs:144
                        }
                        const unsigned long long mask = __ballot(alive);
                        if (alive) live[wv][live_n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = h;
                        live_n += (uint32_t)__popcll(mask);
                        cur = min(cur + 64u, end);
                    }
                    if (live_n == 0) { // the frame has no more paths to start
                        drained = n_idle == 64;
                        break;
                    }
                    wave_lds_fence();
                    m = min(64u, live_n);
                    id = lane < m ? live[wv][live_n - m + lane] : 0u;
                    live_n -= m;
                    wave_lds_fence(); // the slots just read may be written again
                } else {
                    next_chunk();
                    if (cur >= end) { // the pass has no more paths to start
                        drained = n_idle == 64;
                        break;
                    }
                    m = min(64u, end - cur);
                    id = cur + lane;
                    cur += m;
                }
                // the next (up to) 64 camera samples, one per lane, parked in LDS
                const CameraDev cam = load_uniform(&pc->cam);
                const PassParams pp = load_uniform(&pc->pp);
                const bool fresh = lane < m;
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
                    S.cam_pid[wv][k] = (RT && inline_resolve) ? (uint32_t)pixel : id;
                }
                stash_n = (uint32_t)__popcll(made);
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
#ifdef PT_POOL_TIMING
        { long long t = clock64(); tm_refill += t - tm0; tm0 = t; tm_rounds += 1; tm_lanes += __popcll(__ballot(has)); }
#endif
        double ht;
        int32_t id;
        if constexpr (X) {
            unsigned long long unused_nodes = 0, unused_prims = 0;
            // FUSE 3 / 4 (exact mode only): as 1 / 2 for a stream without ENTER / EXIT ops, whose ray never changes frame
            extend_scan_walk<true, FUSE <= 2, FUSE == 2 || FUSE == 4, false>(sv, o, d, time, pkey, depth, 0.001, kF64Max, has, false, ht, id,
                                                           unused_nodes, unused_prims);
        } else {
            extend_list<FUSE == 2>(sv, sv.ops, o, d, time, pkey, depth, 0.001, kF64Max, ht, id); // free lanes walk along
        }
#ifdef PT_POOL_TIMING
        { long long t = clock64(); tm_walk += t - tm0; tm0 = t; }
#endif
        D3 sphere_p = d3(0.0, 0.0, 0.0);
        uint32_t sphere_n = 0;
        if constexpr (kWaveSphere) {
            const int32_t mk = S.leaves[id < 0 ? 0 : (id & 0x0fffffff)].l.mat_kind;
            const bool want = has && id >= 0 && depth < max_depth && (mk == PT_M_ISOTROPIC || mk == PT_M_METAL);
            if (__ballot(want)) wave_unit_sphere(want, pkey, rn, sphere_src[wv], sphere_p, sphere_n);
        }
        if (has) {
            const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
            D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0), c;
            const bool alive = shade_vertex<X, TEX, RAD_RETURN>(sv, S.leaves[li].l, o, d, time, beta, pkey, pid, rn, ht,
                                                                        id, depth, max_depth, flags, lx, ly, lz, no, nd,
                                                                        nbeta, &c, nullptr, kWaveSphere ? &sphere_p : nullptr,
                                                                        sphere_n);
            // the per-path sum of the other forms: the first vertex is stored, later ones are added unless zero
            if (depth == 0) sum = c;
            else if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) sum = sum + c;
            // (a path that ends has no use for its ray any more: the next ray is taken over by every lane that shaded, which
            // spares the copies that keeping the old one for the ended paths would need)
            o = no;
            d = nd;
            beta = nbeta;
            if (alive) {
                depth += 1;
            } else {
                if (RT && inline_resolve) {
                    nan_pixels += blend_pixel(consts, pid, sum);
                } else {
                    lx[pid] = sum.x;
                    ly[pid] = sum.y;
                    lz[pid] = sum.z;
                }
                has = false;
            }
        }
#ifdef PT_POOL_TIMING
        { long long t = clock64(); tm_shade += t - tm0; tm0 = t; }
#endif
    }
#ifdef PT_POOL_TIMING
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        printf("pool timing: rounds %lld lanes/round %.1f cycles/round refill %.0f walk %.0f shade %.0f\n", tm_rounds,
               (double)tm_lanes / tm_rounds, (double)tm_refill / tm_rounds, (double)tm_walk / tm_rounds, (double)tm_shade / tm_rounds);
        const long long *c = shade_clock();
        for (int k = 0; k < kShadePhases; ++k)
            printf("  vertex phase %2d: cycles/round %7.0f  executed in %.3f of rounds  lanes when executed %.1f\n", k,
                   (double)c[3 * k] / tm_rounds, (double)c[3 * k + 1] / tm_rounds, (double)c[3 * k + 2] / (double)(c[3 * k + 1] ? c[3 * k + 1] : 1));
    }
#endif
    // the pass's segment count, for the statistics: 32-bit partial sums spread over the shard slots of bounce row 0
    // (the readers add the slots up in 64 bits), so that the total may pass 2^32
    if (RT && inline_resolve) {
        if (lane == 0 && segments) atomicAdd(load_uniform(&consts->stats), (unsigned long long)segments);
        add_nan_pixels(consts, nan_pixels);
    } else if (lane == 0 && segments) atomicAdd(&counts[blockIdx.x & (kShards - 1u)], segments);
}

// ------------------------------------------------------------------------------------------------------------
// The path pool for scenes that are traced through the SAH tree: one launch per offline pass, a path in one lane's
// registers from camera sample to last vertex, no queues. A lane is free, on an inner node, on a leaf, or waiting
// with a finished traversal. Every round steps the traversing lanes (inner nodes first, then leaves, as in
// k_extend_accel_pool), and once enough lanes wait (or too few still traverse) the waiting lanes shade their vertex
// together and start their next ray; free lanes take new camera samples from the pass's chunk counter.
// ------------------------------------------------------------------------------------------------------------
#ifndef PT_TREE_UV_WAVES
#define PT_TREE_UV_WAVES 3 // waves per SIMD of the variants for noise / image textures, the largest code here. Final scene, Mpaths/s: with the
                           // turbulence evaluated per lane 4: 638, 3: 640, 2: 730; with the wave evaluating it together 2: 994 / 756,
                           // 3: 1 015 / 828 (reference / book semantics); four 45-KiB workgroups do not fit a CU's LDS
#endif
// LDS of the variants for scenes with noise / image textures: the Perlin tables and the scratch of wave_turb7
template <int TEX> struct TreePoolTexLds {};
template <> struct TreePoolTexLds<2> {
    pt_perlin perlin;
    double ps[kWavesPerBlock][64][3];
    double rs[kWavesPerBlock][9][8];
};
// RT: a realtime frame (src/trace.rs:144-146: four of five pixels are skipped, by a draw keyed on seed, frame and
// pixel). The path ids of a frame cover every pixel; a wave evaluates the skip draw of 64 ids at a time with all its
// lanes, parks the ids that survive in LDS, and free lanes take their paths from there.
#ifndef PT_SPH_WAVES
#define PT_SPH_WAVES 4 // waves per SIMD of the variant for plain spheres
#endif
// SPH: the scene is plain spheres alone and `prims` holds their compact records (DevSphere) in leaf order: the leaf
// test and the vertex read the same 64 bytes, and the hit id is the record's index. Same arithmetic on the same
// values as the general form: images are bit-identical.
// QUANT: `nodes` holds the tree in its 32-byte form (AccelNodeQ, planes on consts->grid): two 16-byte reads per lane and
// node visit instead of four, and the slab tests in f32, widened so that they accept whatever the f64 tests accept
// (accel_boxes_qf: the f64 form of these tests measured 4 % slower on the million-sphere render).
template <int TEX, bool MEDIA, bool RT = false, bool SPH = false, bool QUANT = false>
__global__ __launch_bounds__(kBlock, TEX == 2 ? PT_TREE_UV_WAVES : (SPH ? PT_SPH_WAVES : PT_SHADE_WAVES)) void k_tree_pool(PoolConsts consts_by_value, SceneView sv, const AccelNode *nodes, const DevOp *prims,
                                                                      const PoolConsts *consts_in_memory, uint32_t n, uint32_t max_depth,
                                                                      double *lx, double *ly, double *lz, uint32_t *counts,
                                                                      uint32_t flags, uint32_t tuning, uint32_t chunk_arg) {
    const PoolConsts *consts = pool_consts_of<RT>(consts_in_memory);
    bool inline_resolve = false; // as in k_path_pool
    if constexpr (RT) inline_resolve = load_uniform(&consts->resolve_inline) != 0u && (chunk_arg >> 31) != 0u;
    uint32_t nan_pixels = 0;
    static_assert(!QUANT || SPH, "the 32-byte nodes are read by the variant for plain spheres only");
    static_assert(!SPH || (TEX <= 1 && !MEDIA), "compact records: constant colours and checkers, no media");
    const uint32_t chunk = chunk_arg & 0x7fffffffu;
    const bool static_chunks = (chunk_arg >> 31) != 0u; // RT: one chunk per wave, dealt by wave index
    bool started = false;
    __shared__ int32_t stack[kAccelMaxDepth][kBlock]; // [depth][thread]: conflict-free
    __shared__ uint32_t live[RT ? kWavesPerBlock : 1][64]; // RT: surviving path ids of the wave, not yet taken
    uint32_t live_n = 0;
    const uint32_t refill_at = min(max(tuning & 0xffu, 1u), 64u), inner_min = (tuning >> 8) & 0xffu, shade_min = (tuning >> 16) & 0xffu,
                   trav_min = tuning >> 24;
    constexpr int32_t kDone = INT32_MIN; // never a leaf reference
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ TreePoolTexLds<TEX> T;
    // The variant for noise / image textures runs a hundred registers over its budget: its paths' radiance sums, which a
    // vertex touches once, live in LDS (6 KiB per workgroup) instead of six registers per lane.
    constexpr bool kSumInLds = TEX == 2;
    __shared__ double sum_lds[kSumInLds ? 3 : 1][kSumInLds ? kBlock : 1];
    if constexpr (TEX == 2) {
        if (sv.perlin) {
            const uint32_t *src = (const uint32_t *)sv.perlin;
            for (uint32_t k = threadIdx.x; k < sizeof(pt_perlin) / 4u; k += kBlock) ((uint32_t *)&T.perlin)[k] = src[k];
        }
        __syncthreads(); // the only barrier of the kernel
    }
    uint32_t cur = 0, end = 0;
    bool exhausted = false;
    bool has = false;
    D3 o = d3(0, 0, 0), d = d3(0, 0, 1), inv = o, beta = o, sum = o;
    double time = 0.0, closest = 0.0;
    uint64_t pkey = 0;
    uint32_t pid = 0, rn = 0, depth = 0, segments = 0;
    int32_t best = -1, node = kDone;
    int sp = 0;
#ifdef PT_POOL_TIMING
    long long tm_refill = 0, tm_inner = 0, tm_leaf = 0, tm_shade = 0, tm_rounds = 0, tm_shades = 0, tm_lanes = 0, tm0 = clock64();
    long long tm_leaf_steps = 0, tm_leaf_box = 0, tm_leaf_sph = 0, tm_leaf_med = 0, tm_leaf_chain = 0, tm_leaf_all = 0, tm_inner_steps = 0, tm_inner_lanes = 0;
#define PT_TICK(acc) { long long t_ = clock64(); acc += t_ - tm0; tm0 = t_; }
    for (int k = lane; k < 3 * kShadePhases + 1; k += 64) shade_clock()[k] = 0;
#else
#define PT_TICK(acc)
#endif
    QuantRayF qr{};
    float closest_up = 0.0f; // the closest hit so far as an f32 that is not below it
    auto start_ray = [&]() { // (o, d) hold a new ray
        int32_t first_node = 0;
        sp = 0;
        if constexpr (QUANT) {
            const PoolConsts *pc = consts;
            asm volatile("" : "+s"(pc)); // keeps the loads in this block
            first_node = load_uniform(&pc->start_node);            // the root, or the leaf the root holds ...
            stack[0][threadIdx.x] = load_uniform(&pc->start_push); // ... with the rest of the tree behind it
            sp = first_node < 0 ? 1 : 0;
            qr = quant_ray_f(load_uniform(&pc->grid), o, d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z)));
            closest_up = __builtin_inff();
        } else {
            inv = d3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
        }
        closest = kF64Max;
        best = -1;
        // rays with a NaN or infinite component are not traced (see k_extend_accel)
        const double probe = (o.x + o.y + o.z) + (d.x + d.y + d.z);
        node = (probe == probe && fabs(probe) <= kF64Max) ? first_node : kDone;
    };
    // pixel of the h-th path id of the pass
    auto pixel_of = [&](const PassParams &pp, uint32_t h, uint32_t &i, uint32_t &jy, uint32_t &smp) -> uint64_t {
        const uint32_t p = h % pp.cpix;
        smp = h / pp.cpix;
        i = pp.cx0 + p % pp.cw;
        jy = pp.cy0 + (p / pp.cw) * pp.row_step;
        return (uint64_t)jy * pp.width + i;
    };
    // the calling lanes start path id h: camera sample (src/trace.rs:153-156, Camera::get_ray src/camera.rs:53-59)
    auto take_path = [&](uint32_t h) {
        const PoolConsts *pc = consts;
        asm volatile("" : "+s"(pc)); // keeps the loads below in this block
        const CameraDev cam = load_uniform(&pc->cam);
        const PassParams pp = load_uniform(&pc->pp);
        uint32_t i, jy, smp;
        const uint64_t pixel = pixel_of(pp, h, i, jy, smp);
        pid = (RT && inline_resolve) ? (uint32_t)pixel : h;
        Rng g{path_key(pp.seed, pp.frame, pixel, pp.s_base + smp), 0};
        double random = g.next();
        const double u = ((double)i + random) / (double)pp.width;
        random = g.next();
        const double v = ((double)jy + random) / (double)pp.height;
        const D3 rd = random_in_unit_disk(g) * cam.lens_radius;
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
    };
    // the wave's next chunk of path ids, when the current one is used up
    auto next_chunk = [&]() {
        if (cur >= end && !exhausted) {
            uint32_t first = 0;
            if (RT && static_chunks) { // the launch has a chunk for every wave: wave k takes chunk k, no counter (see k_path_pool)
                first = started ? n : min((blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * chunk, n);
                started = true;
            } else if (lane == 0) first = atomicAdd(&counts[(size_t)(max_depth + 1) * kShards], chunk);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            exhausted = first >= n;
            cur = min(first, n);
            end = min(cur + chunk, n);
        }
    };
    for (;;) {
        unsigned long long idle = __ballot(!has);
        uint32_t n_idle = (uint32_t)__popcll(idle);
        if constexpr (RT) {
            bool drained = false;
            while (n_idle >= refill_at) {
                if (live_n == 0) { // the skip draws of the next 64 ids, by all lanes
                    next_chunk();
                    if (cur >= end) {
                        drained = n_idle == 64;
                        break;
                    }
                    const PoolConsts *pc = consts;
                    asm volatile("" : "+s"(pc));
                    const PassParams pp = load_uniform(&pc->pp);
                    const uint32_t h = cur + lane;
                    bool alive = false;
                    if (h < end) {
                        uint32_t i, jy, smp;
                        const uint64_t pixel = pixel_of(pp, h, i, jy, smp);
                        alive = !(pixel_skip_draw(pp.seed, pp.frame, pixel) < 0.8); // src/trace.rs:144
                    }
                    const unsigned long long mask = __ballot(alive);
                    if (alive) live[threadIdx.x >> 6][__popcll(mask & ((1ull << lane) - 1ull))] = h;
                    live_n = (uint32_t)__popcll(mask);
                    cur = min(cur + 64u, end);
                    wave_lds_fence();
                    if (live_n == 0) continue;
                }
                const uint32_t take = min(n_idle, live_n);
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!has && rank < take) take_path(live[threadIdx.x >> 6][live_n - 1u - rank]);
                live_n -= take;
                wave_lds_fence(); // the slots just read may be written again
                idle = __ballot(!has);
                n_idle = (uint32_t)__popcll(idle);
            }
            if (drained) break; // nothing in flight, nothing left to start
        } else if (n_idle >= refill_at) {
            next_chunk();
            if (cur < end) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!has && cur + rank < end) take_path(cur + rank);
                cur = min(cur + n_idle, end);
            } else if (n_idle == 64) {
                break; // nothing in flight, nothing left to start
            }
        }
        PT_TICK(tm_refill)
        // traversal: inner nodes for as long as at least `inner_min` lanes are on one, then the leaves
        for (;;) {
            if (has && node >= 0) {
                bool h0, h1, first0;
                int32_t c0, c1;
                if constexpr (QUANT) {
                    const AccelNodeQWords nq = ((const AccelNodeQWords *)nodes)[node];
                    float t0, t1;
                    accel_boxes_qf(nq, qr, 0.000999f, closest_up, h0, h1, t0, t1); // (below the 0.001 of a render's rays)
                    c0 = (int32_t)nq.b.z;
                    c1 = (int32_t)nq.b.w;
                    first0 = t0 <= t1;
                } else {
                    double t0, t1;
                    const AccelNode &nd = nodes[node];
                    h0 = accel_box(nd.lo[0], nd.hi[0], o, inv, 0.001, closest, t0);
                    h1 = accel_box(nd.lo[1], nd.hi[1], o, inv, 0.001, closest, t1);
                    first0 = t0 <= t1;
                    c0 = nd.child[0];
                    c1 = nd.child[1];
                }
                if (h0 && h1) {
                    stack[sp++][threadIdx.x] = first0 ? c1 : c0;
                    node = first0 ? c0 : c1;
                } else if (h0 || h1) {
                    node = h0 ? c0 : c1;
                } else {
                    node = sp ? stack[--sp][threadIdx.x] : kDone;
                }
            }
            const uint32_t n_inner = (uint32_t)__popcll(__ballot(has && node >= 0));
#ifdef PT_POOL_TIMING
            tm_inner_steps += 1;
            tm_inner_lanes += n_inner;
#endif
            if (n_inner == 0 || n_inner < inner_min) break;
        }
        PT_TICK(tm_inner)
        if (has && node < 0 && node != kDone) {
            const int32_t ref = ~node;
            const int32_t first = ref >> 3, count = ref & 7;
            for (int32_t k = first; k < first + count; ++k) {
                double t = 0.0;
                int sub = 0;
                if constexpr (SPH) {
                    const DevSphere *sp = (const DevSphere *)prims + k;
                    // (the test reads the radius only squared: its sign bit, part of the material kind, drops out)
                    if (sphere_t<false>(d3(sp->c[0], sp->c[1], sp->c[2]), sp->radius, o, d, 0.001, closest, t, sub)) {
                        closest = t;
                        best = k | (sub << 28);
                        if constexpr (QUANT) closest_up = f32_not_below(t);
                    }
                } else {
                    const DevOp op = prims[k];
#ifdef PT_POOL_TIMING
                    { // lanes per primitive kind in this step of the leaf loop
                        const int kd = op.kind & 0xff;
                        tm_leaf_steps += 1;
                        tm_leaf_box += __popcll(__ballot(kd == OP_BOXPRIM));
                        tm_leaf_sph += __popcll(__ballot(kd == OP_SPHERE || kd == OP_MSPHERE));
                        tm_leaf_med += __popcll(__ballot(kd == OP_MEDIUM));
                        tm_leaf_chain += __popcll(__ballot(((op.kind >> 16) & 0xff) != 0));
                        tm_leaf_all += __popcll(__ballot(true));
                    }
#endif
                    bool hit;
                    if (MEDIA && (op.kind & 0xff) == OP_MEDIUM)
                        hit = medium_hit<false>(sv, op, o, d, inv, time, 0.001, closest, pkey, depth, t);
                    else hit = prim_hit<false>(sv, op, o, d, inv, time, 0.001, closest, t, sub);
                    if (hit) {
                        closest = t;
                        best = op.leaf | (sub << 28);
                    }
                }
            }
            node = sp ? stack[--sp][threadIdx.x] : kDone;
        }
        PT_TICK(tm_leaf)
        // vertices: when enough lanes wait with a finished traversal, or too few are still traversing to go on
        const bool ready = has && node == kDone;
        const uint32_t n_ready = (uint32_t)__popcll(__ballot(ready));
        const uint32_t n_trav = (uint32_t)__popcll(__ballot(has && node != kDone));
#ifdef PT_POOL_TIMING
        tm_rounds += 1;
        if (n_ready && (n_ready >= shade_min || n_trav <= trav_min)) { tm_shades += 1; tm_lanes += n_ready; }
#endif
        if (n_ready && (n_ready >= shade_min || n_trav <= trav_min)) {
            segments += n_ready;
            const int32_t li = best < 0 ? 0 : (best & 0x0fffffff);
            double turbulence = 0.0;
            bool noise_hit = false;
            if constexpr (TEX == 2) { // hits on noise-textured leaves: the wave evaluates their turbulence together
                noise_hit = ready && best >= 0 && (sv.leaves[li].tex_flags & kLeafTexNoise) != 0;
                if (__ballot(noise_hit)) {
                    D3 p = o;
                    if (noise_hit) p = finalize_hit<false, TEX>(sv, sv.leaves[li], o, d, time, closest, (best >> 28) & 7).p;
                    const uint32_t wv = threadIdx.x >> 6;
                    turbulence = wave_turb7(&T.perlin, T.ps[wv], T.rs[wv], noise_hit, p);
                }
            }
            if (ready) {
                D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0), c;
                bool alive;
                if constexpr (SPH) {
                    // the leaf record the vertex code reads, rebuilt in registers from the compact record
                    const DevSphere sp = ((const DevSphere *)prims)[li];
                    DevLeaf lf;
                    lf.kind = OP_SPHERE;
                    lf.axis = 0;
                    lf.mat_kind = 1 + (int32_t)((uint32_t)__double2hiint(sp.radius) >> 31) +
                                  2 * (int32_t)((uint32_t)__double2hiint(sp.param) >> 31);
                    lf.texture = (TEX != 0 && sp.albedo[1] != sp.albedo[1]) ? (int32_t)sp.albedo[0] : -1; // (see DevSphere)
                    lf.full_start = lf.full_count = 0;
                    lf.material = lf.op = lf.prim[0] = 0;
                    lf.tex_flags = 0;
                    lf.g[0] = sp.c[0]; lf.g[1] = sp.c[1]; lf.g[2] = sp.c[2];
                    lf.g[3] = fabs(sp.radius);
                    lf.albedo[0] = sp.albedo[0]; lf.albedo[1] = sp.albedo[1]; lf.albedo[2] = sp.albedo[2];
                    lf.emissive = lf.ref_idx = lf.fuzz = fabs(sp.param);
                    alive = shade_vertex<false, TEX, RAD_RETURN>(sv, lf, o, d, time, beta, pkey, pid, rn, closest, best, depth,
                                                                    max_depth, flags, lx, ly, lz, no, nd, nbeta, &c, nullptr);
                } else {
                    alive = shade_vertex<false, TEX, RAD_RETURN>(sv, sv.leaves[li], o, d, time, beta, pkey, pid, rn, closest, best,
                                                                    depth, max_depth, flags, lx, ly, lz, no, nd, nbeta, &c,
                                                                    noise_hit ? &turbulence : nullptr);
                }
                if constexpr (kSumInLds) { // the path's radiance sum is parked in LDS between its vertices
                    if (depth == 0) sum = c;
                    else {
                        sum = d3(sum_lds[0][threadIdx.x], sum_lds[1][threadIdx.x], sum_lds[2][threadIdx.x]);
                        if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) sum = sum + c;
                    }
                    sum_lds[0][threadIdx.x] = sum.x;
                    sum_lds[1][threadIdx.x] = sum.y;
                    sum_lds[2][threadIdx.x] = sum.z;
                } else {
                    if (depth == 0) sum = c; // the per-path sum of the other forms
                    else if (c.x != 0.0 || c.y != 0.0 || c.z != 0.0) sum = sum + c;
                }
                o = no; // (taken over by every lane that shaded: see k_path_pool)
                d = nd;
                beta = nbeta;
                if (alive) {
                    depth += 1;
                    start_ray();
                } else {
                    if (RT && inline_resolve) {
                        nan_pixels += blend_pixel(consts, pid, sum);
                    } else {
                        lx[pid] = sum.x;
                        ly[pid] = sum.y;
                        lz[pid] = sum.z;
                    }
                    has = false;
                }
            }
        }
        PT_TICK(tm_shade)
    }
#ifdef PT_POOL_TIMING
    if (blockIdx.x == 7 && threadIdx.x == 0)
        printf("tree pool timing: rounds %lld shade rounds %lld lanes/shade %.1f cycles/round refill %.0f inner %.0f leaf %.0f shade %.0f (per shade round %.0f)\n",
               tm_rounds, tm_shades, (double)tm_lanes / (double)(tm_shades ? tm_shades : 1), (double)tm_refill / tm_rounds, (double)tm_inner / tm_rounds,
               (double)tm_leaf / tm_rounds, (double)tm_shade / tm_rounds, (double)tm_shade / (double)(tm_shades ? tm_shades : 1));
    if (blockIdx.x == 7 && threadIdx.x == 0)
        printf("  inner steps/round %.2f lanes left on inner nodes after a step %.1f | leaf-loop steps/round %.2f lanes/step %.1f: box %.1f sphere %.1f medium %.1f (with a chain %.1f)\n",
               (double)tm_inner_steps / tm_rounds, (double)tm_inner_lanes / (double)(tm_inner_steps ? tm_inner_steps : 1),
               (double)tm_leaf_steps / tm_rounds, (double)tm_leaf_all / (double)(tm_leaf_steps ? tm_leaf_steps : 1),
               (double)tm_leaf_box / (double)(tm_leaf_steps ? tm_leaf_steps : 1), (double)tm_leaf_sph / (double)(tm_leaf_steps ? tm_leaf_steps : 1),
               (double)tm_leaf_med / (double)(tm_leaf_steps ? tm_leaf_steps : 1), (double)tm_leaf_chain / (double)(tm_leaf_steps ? tm_leaf_steps : 1));
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        const long long *c = shade_clock();
        for (int k = 0; k < kShadePhases; ++k)
            printf("  vertex phase %2d: cycles/shade round %7.0f  executed in %.3f of shade rounds  lanes when executed %.1f\n", k,
                   (double)c[3 * k] / tm_shades, (double)c[3 * k + 1] / tm_shades, (double)c[3 * k + 2] / (double)(c[3 * k + 1] ? c[3 * k + 1] : 1));
    }
#endif
    if (RT && inline_resolve) { // as in k_path_pool
        if (lane == 0 && segments) atomicAdd(load_uniform(&consts->stats), (unsigned long long)segments);
        add_nan_pixels(consts, nan_pixels);
    } else if (lane == 0 && segments) atomicAdd(&counts[blockIdx.x & (kShards - 1u)], segments);
}
