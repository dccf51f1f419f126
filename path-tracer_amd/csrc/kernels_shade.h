// One path vertex (the body of `color`, src/trace.rs:198-234) and the queue-based pass forms built on it: a shade launch per bounce, and persistent waves that trace and shade.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

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

#ifdef PT_POOL_TIMING
// Diagnostic build: cycles, executions and active lanes of the parts of a vertex, per wave, kept in LDS (every active
// lane stores the same value). Phase 0 closes at the call's entry, so it holds the time between vertices.
constexpr int kShadePhases = 12;
__device__ __forceinline__ long long *shade_clock() {
    __shared__ long long clk[kWavesPerBlock][3 * kShadePhases + 1];
    return clk[threadIdx.x >> 6];
}
#define PT_SHADE_TICK(k)                                                        \
    {                                                                           \
        long long *c_ = shade_clock();                                          \
        const long long t_ = clock64();                                         \
        c_[3 * (k)] += t_ - c_[3 * kShadePhases];                               \
        c_[3 * (k) + 1] += 1;                                                   \
        c_[3 * (k) + 2] += __popcll(__ballot(true));                            \
        c_[3 * kShadePhases] = t_;                                              \
    }
#else
#define PT_SHADE_TICK(k)
#endif

// The light shape each calling lane aims at (`which`, for the lanes that `want` one): the list is a handful of records, so
// each is read once for the wave at a uniform address and the lanes that drew it keep it.
__device__ __forceinline__ DevLight wave_light(const SceneView &sv, int which, bool want) {
    DevLight l = load_light_uniform(sv.lights);
    for (int k = 1; k < sv.n_lights; ++k) {
        if (__ballot(want && which == k)) {
            const DevLight lk = load_light_uniform(sv.lights + k);
            if (which == k) l = lk;
        }
    }
    return l;
}

// One path vertex: the body of `color` (src/trace.rs:198-234) for a ray whose closest hit is (ht, id), id < 0 for a
// miss. `lf` is the hit primitive's leaf record (fields are fetched where they are used; any record when id < 0).
// Adds the vertex's radiance to the path's accumulator and, if the path goes on, returns true with the next ray,
// throughput and RNG draw count.
// `turbulence`: the Perlin turbulence at this hit when the caller has evaluated it already (wave_turb7), else null.
// RAD_RETURN: the vertex's radiance (every branch adds at most once) is handed back in `vertex_radiance` and nothing
// is written (the caller keeps the path's sum).
// `pre_sphere`: the point random_in_unit_sphere gives at this vertex and the draw count behind it (`pre_n`), when the
// caller has made them already (wave_unit_sphere) for every Metal and Isotropic hit below the depth cap; else null.
enum RadianceMode { RAD_ADD = 0, RAD_RETURN = 2 };
template <bool X, int TEX, int FIRST = RAD_ADD>
__device__ __forceinline__ bool shade_vertex(const SceneView &sv, const DevLeaf &lf, D3 o, D3 d, double time, D3 beta,
                                             uint64_t pkey, uint32_t pid, uint32_t &rn, double ht, int32_t id,
                                             uint32_t depth, uint32_t max_depth, uint32_t flags, double *lx, double *ly,
                                             double *lz, D3 &no, D3 &nd, D3 &nbeta, D3 *vertex_radiance = nullptr,
                                             const double *turbulence = nullptr, const D3 *pre_sphere = nullptr,
                                             uint32_t pre_n = 0) {
    bool alive = false;
    D3 first = d3(0.0, 0.0, 0.0);
    auto add_radiance = [&](double *ax, double *ay, double *az, uint32_t path, D3 c) {
        if constexpr (FIRST != RAD_ADD) first = c;
        else ::add_radiance(ax, ay, az, path, c);
    };
    PT_SHADE_TICK(0)
    if (id < 0) {
        // miss: colour is black (src/trace.rs:227); the product still carries a NaN throughput through
        add_radiance(lx, ly, lz, pid, beta * 0.0);
    } else {
        Rng g{pkey, rn};
        auto unit_sphere_point = [&]() -> D3 { // src/material.rs:21-31
            if (pre_sphere) {
                g.n = pre_n;
                return *pre_sphere;
            }
            return random_in_unit_sphere(g);
        };
        const HitRec rec = finalize_hit<X, TEX>(sv, lf, o, d, time, ht, (id >> 28) & 7);
        PT_SHADE_TICK(1)
        const D3 emissive = emitted<X, TEX>(sv, lf, d, rec, turbulence);
        PT_SHADE_TICK(2)
        bool done = true, dead = false;
        if (depth < max_depth) {
            if (lf.mat_kind == PT_M_DIELECTRIC) { // src/material.rs:177-212
                D3 outward_normal;
                D3 reflected = reflect<X>(d, rec.n);
                double ni_over_nt, reflect_prob, cosine;
                D3 refracted = d3(0.0, 0.0, 0.0);
                const double d_dot_n = M<X>::dot(d, rec.n);
                if constexpr (X) {
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
                } else {
                    // the same with 1 / |d| formed once: the reference divides by |d| for the cosine and normalises d
                    // again inside refract (src/material.rs:55,189,194)
                    const bool outside = !(d_dot_n > 0.0);
                    const double inv_len = fast_rsqrt(M<X>::squared_length(d));
                    const double inv_idx = fast_rcp(lf.ref_idx);
                    outward_normal = outside ? rec.n : -rec.n;
                    ni_over_nt = outside ? inv_idx : lf.ref_idx;
                    cosine = (outside ? -d_dot_n : lf.ref_idx * d_dot_n) * inv_len;
                    const D3 uv = d * inv_len;
                    const double dt = (outside ? d_dot_n : -d_dot_n) * inv_len; // uv . outward_normal
                    const double discriminant = 1.0 - ni_over_nt * ni_over_nt * (1.0 - dt * dt);
                    reflect_prob = 1.0;
                    if (discriminant > 0.0) {
                        refracted = ni_over_nt * (uv - outward_normal * dt) - outward_normal * fast_sqrt(discriminant);
                        reflect_prob = schlick<X>(cosine, lf.ref_idx);
                    }
                }
                nd = (g.next() < reflect_prob) ? reflected : refracted;
                no = rec.p;
                nbeta = beta * d3(1.0, 1.0, 1.0); // specular: albedo * color(next), no emission (trace.rs:209-211)
                done = false;
                PT_SHADE_TICK(3)
            } else if (lf.mat_kind == PT_M_METAL) { // src/material.rs:230-241
                D3 reflected = reflect<X>(M<X>::unit_vector(d), rec.n);
                nd = reflected + lf.fuzz * unit_sphere_point();
                no = rec.p;
                nbeta = beta * d3(lf.albedo[0], lf.albedo[1], lf.albedo[2]);
                done = false;
                PT_SHADE_TICK(4)
            } else if (lf.mat_kind == PT_M_LAMBERTIAN || lf.mat_kind == PT_M_ISOTROPIC) {
                const bool lambert = lf.mat_kind == PT_M_LAMBERTIAN;
                // scatter(): src/material.rs:289-301 (Lambertian), 347-351 (Isotropic draws a sphere sample it never uses)
                // With PT_RENDER_BOOK_SEMANTICS that sample is the scattered direction and the pdf is the
                // uniform sphere's, 1/(4 pi) (unit_sphere_pdf, src/material.rs:46-48).
                const bool book = (flags & PT_RENDER_BOOK_SEMANTICS) != 0 && !lambert;
                D3 iso_dir = d3(0.0, 0.0, 0.0);
                if (!lambert) {
                    const D3 sample = unit_sphere_point();
                    if (book) iso_dir = sample;
                }
                const double iso_pdf = book ? 0.25 * kFrac1Pi : 0.0;
                const D3 albedo = leaf_texture<X, TEX>(sv, lf, rec, turbulence);
                add_radiance(lx, ly, lz, pid, beta * emissive);
                Onb uvw; // exact mode: the basis of CosinePDF::new(normal), src/material.rs:363-367
                if constexpr (X)
                    if (lambert) uvw = onb_from_w<X>(rec.n);
                // MixturePDF::generate (src/material.rs:427-433), then HitableList::random (src/hitable.rs:74-77) or
                // the material's own pdf. Every sampler a lane can end up in takes its next two draws (or none):
                // their bits are made once, by all lanes together, before the lanes part ways.
                const bool to_light = g.next() < 0.5;
                int which = 0;
                if (to_light) which = (int)((double)sv.n_lights * g.next());
                const uint32_t u1 = g.peek_u32(1), u2 = g.peek_u32(2);
                D3 dir;
                PT_SHADE_TICK(5)
                if constexpr (X) {
                    const DevLight l = wave_light(sv, which, to_light);
                    if (to_light) dir = light_random<X>(l, u1, u2, g, rec.p);
                    else dir = lambert ? onb_local(uvw, random_cosine_direction<X>(u1, u2, g)) : iso_dir;
                } else {
                    // The cone towards a sphere light (src/sphere.rs:16-25,96-101) and the cosine lobe about the normal
                    // (src/material.rs:10-19) are the same construction: a basis about an axis, z from the second
                    // draw, the azimuth from the first. The lanes of both run it together instead of one group after
                    // the other; what differs is selected per lane.
                    // The light shape a lane aims at comes through the scalar cache: the list is a handful of records,
                    // so each is read once for the wave (a uniform address) and the lanes that drew it keep it. A load
                    // at the lane's own index is a vector load whose round trip sits right in front of the sampling.
                    const DevLight l = wave_light(sv, which, to_light);
                    const bool to_sphere = to_light && l.kind == LIGHT_SPHERE;
                    if (to_sphere || (!to_light && lambert)) {
                        g.n += 2;
                        const double r1 = bits_to_f64(u1), r2 = bits_to_f64(u2);
                        const D3 axis = to_sphere ? d3(l.d[0], l.d[1], l.d[2]) - rec.p : rec.n;
                        const double radius2 = l.d[3] * l.d[3];
                        // 1 / |axis| serves twice: squared, for the cone's opening, and as the basis' normalisation
                        const double inv_axis = weight_rsqrt(M<X>::squared_length(axis));
                        // sphere: cos_theta_max = sqrt(1 - r^2 / |axis|^2), z = 1 + r2 (cos_theta_max - 1); lobe: z = sqrt(1 - r2)
                        const double q = to_sphere ? radius2 * (inv_axis * inv_axis) : r2;
                        const double root = weight_sqrt(1.0 - q);
                        const double z = to_sphere ? __builtin_fma(r2, root - 1.0, 1.0) : root;
                        // sphere: sqrt(1 - z^2); lobe: sqrt(r2)
                        const double planar = weight_sqrt(to_sphere ? __builtin_fma(-z, z, 1.0) : r2);
                        double sn, cs;
                        fast_sincos_2pi(r1, sn, cs);
                        dir = onb_local_fma(onb_from_w_scaled(axis, inv_axis), d3(cs * planar, sn * planar, z));
                        PT_SHADE_TICK(6)
                    } else if (to_light) {
                        dir = light_random<X>(l, u1, u2, g, rec.p); // rectangle, or a shape without random()
                        PT_SHADE_TICK(7)
                    } else {
                        dir = iso_dir;
                    }
                }
                PT_SHADE_TICK(8)
                // MixturePDF::value, src/material.rs:424-426
                double len2 = 0.0, inv_len = 0.0;
                D3 unit_dir;
                if constexpr (X) unit_dir = M<X>::unit_vector(dir);
                else {
                    len2 = M<X>::squared_length(dir);
                    inv_len = weight_rsqrt(len2);
                    unit_dir = dir * inv_len;
                }
                double own = iso_pdf;
                if (lambert) { // CosinePDF::value, src/material.rs:375-382
                    // (fast mode: the basis' w is the unit normal itself up to its last bit)
                    double cosine = X ? M<X>::dot(unit_dir, uvw.w) : M<X>::dot(unit_dir, rec.n);
                    own = cosine > 0.0 ? cosine * kFrac1Pi : 0.0;
                }
                PT_SHADE_TICK(9)
                const double pdf_val = 0.5 * lights_pdf_value<X>(sv, rec.p, dir, len2, inv_len) + 0.5 * own;
                PT_SHADE_TICK(10)
                double spdf = iso_pdf;
                if (lambert) { // Lambertian::scattering_pdf, src/material.rs:268-275
                    double cosine = M<X>::dot(rec.n, unit_dir);
                    spdf = cosine < 0.0 ? 0.0 : cosine * kFrac1Pi;
                }
                // colour = albedo * spdf * color(next) / pdf_val + emissive (src/trace.rs:217-221)
                nbeta = W<X>::divs(beta * (albedo * spdf), pdf_val);
                nd = dir;
                if constexpr (!X) {
                    // A Lambertian direction below the surface has scattering_pdf 0: the reference goes on with a
                    // throughput of 0 * x / pdf = 0 (the mixture pdf is positive there: the direction came from a
                    // light). Such a path can only add 0 * (finite radiance) from here on; unless the scene has a
                    // source of NaN weights (SceneView::zero_can_turn_nan) fast mode stops it. A NaN or zero pdf
                    // makes the weight NaN, as it does in the reference, and the path goes on.
                    const bool zero = nbeta.x == 0.0 && nbeta.y == 0.0 && nbeta.z == 0.0;
                    dead = lambert && spdf == 0.0 && zero && !sv.zero_can_turn_nan;
                }
                no = rec.p;
                done = false;
                PT_SHADE_TICK(11)
            }
        }
        if (done) add_radiance(lx, ly, lz, pid, beta * emissive); // src/trace.rs:225
        alive = !done && !dead;
        rn = g.n;
    }
    if constexpr (FIRST == RAD_RETURN) *vertex_radiance = first;
    return alive;
}

// The kernels that trace AND shade (FUSE = 1, or 2 with media): the scene is a short box-free op list, walked right
// before the vertex is shaded, so no extend launch, no re-read of the ray and no hit round trip through HBM. The op
// list is passed in sv.ops / sv.n_traversal_ops.
// A leaf record padded to 200 B: with a stride of 50 dwords the same field of different leaves spreads over 16 LDS
// banks instead of 2.
struct LdsLeaf {
    DevLeaf l;
    double pad;
};
constexpr int kLdsLeaves = 32; // the fused kernels keep the whole leaf table in LDS

// LDS of one workgroup. Kept as ONE object: with a second __shared__ object beside a buffer that direct-to-LDS loads
// fill, the compiler waits for those loads before every LDS read.
struct ShadeLds {
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

// Shade stage on its own: hits come from the extend kernel through hit_t / hit_id (scenes traced through a tree in
// realtime frames and in exact mode). Survivors are compacted into the output queue: wave64 ballot + popcount, wave
// totals in LDS, one atomic per workgroup and shard.
template <bool X, int TEX>
__global__ __launch_bounds__(kBlock, X ? 1 : PT_SHADE_WAVES) void k_shade(Queue qi, Queue qo, QueueShape sh, const uint32_t *n_in,
                                                  uint32_t *n_out, SceneView sv, const double *hit_t, const int32_t *hit_id,
                                                  double *lx, double *ly, double *lz,
                                                  uint32_t depth, uint32_t max_depth, uint32_t flags) {
    __shared__ uint32_t lds[kWavesPerBlock + 1];
    const uint32_t n_virtual = ((max_shard_count(n_in, sh.shards) + kBlock - 1) / kBlock) << sh.shift;
    for (uint32_t v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        const uint32_t s_id = v & (sh.shards - 1);
        const uint32_t ns = shard_count(n_in, s_id);
        const uint32_t row0 = (v >> sh.shift) * kBlock;
        if (row0 >= ns) continue; // uniform for the workgroup
        bool alive = false;
        D3 no = d3(0, 0, 0), nd = d3(0, 0, 0), nbeta = d3(0, 0, 0);
        double time = 0.0;
        uint32_t pid = 0, rn = 0;
        uint64_t pkey = 0;
        const uint32_t local = row0 + threadIdx.x;
        const uint32_t i = s_id * sh.shard_cap + local;
        const uint32_t n = s_id * sh.shard_cap + ns; // first slot past this shard's entries
        if (i < n) {
            const D3 o = d3(qi.ox(i), qi.oy(i), qi.oz(i));
            const D3 d = d3(qi.dx(i), qi.dy(i), qi.dz(i));
            time = qi.tm(i);
            pid = qi.pid(i);
            const int32_t id = hit_id[i];
            const double ht = hit_t[i];
            const D3 beta = d3(qi.bx(i), qi.by(i), qi.bz(i));
            rn = qi.rn(i);
            pkey = qi.key(i);
            const int32_t li = id < 0 ? 0 : (id & 0x0fffffff);
            alive = shade_vertex<X, TEX>(sv, sv.leaves[li], o, d, time, beta, pkey, pid, rn, ht, id, depth, max_depth, flags,
                                         lx, ly, lz, no, nd, nbeta);
        }
        const uint32_t slot = s_id * sh.shard_cap + compact_slot(alive, &n_out[s_id], lds);
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
// Persistent form of the fused pipeline: ONE launch traces a whole pass. The pass's paths are dealt to the resident
// waves in contiguous runs of queue tiles; a wave then carries its own paths through every bounce, compacting the
// survivors into the same run of tiles of the other queue. A wave shares nothing with any other wave: no atomics on
// shared counters, no barriers, no launch per bounce, and the tile prefetch runs on across bounces. The price is one
// partly filled tile per wave and bounce, a few per cent of the tile steps. Which wave carries a path has no effect
// on the path (its RNG is keyed by pixel and sample), so images are bit-identical to the launch-per-bounce form.
// ------------------------------------------------------------------------------------------------------------
template <int TEX, int FUSE>
__global__ __launch_bounds__(kBlock, PT_SHADE_WAVES) void k_trace_persistent(Queue qa, Queue qb, uint32_t *counts,
                                                                             SceneView sv, double *lx, double *ly,
                                                                             double *lz, uint32_t max_depth,
                                                                             uint32_t flags) {
    __shared__ ShadeLds S;
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
                alive = shade_vertex<false, TEX>(sv, S.leaves[li].l, o, d, time, beta, pkey, pid, rn, ht, id, depth,
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
