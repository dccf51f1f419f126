// Extend stage: the threaded walk of the reference's tree (per lane and wave-uniform), the fused primitive-list walk and the SAH-tree traversal of fast mode.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

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
// MEDIA: 0 no MEDIUM op, 1 media whose boundary is one primitive, 2 media with a list or tree as boundary.
template <bool X, bool FRAMES, int MEDIA>
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
                if (medium_hit<X, MEDIA == 2>(sv, *op, o, d, inv, time, a.t_min, closest, pkey, a.depth, t)) {
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
                hit = medium_hit<X, false, true>(sv, op, o, d, inv, time, t_min, closest, pkey, depth, t); // (the op is the wave's)
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
// Closest hit over the fused kernels' op list (see pt_scene_upload): no BOX ops, and a primitive's transform chain
// is spelled out as XFORM ops in front of it. Every lane of the wave tests every primitive, in stream order. Op
// records come through the scalar cache, one iteration ahead of their use, so a wave never waits for the record it
// is about to test. Must be called by all 64 lanes of a wave. Fast mode only.
constexpr int kOpFramed = 1 << 24; // fused lists: the primitive is tested against the frame ray set up by XFORM ops
constexpr int kOpFirst = 1 << 25;  // fused lists: this XFORM op starts a chain (from the world ray)

template <bool MEDIA> struct ListWalk {
    const SceneView &sv;
    D3 o, d, inv; // world ray
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
            hit = medium_hit<false, false, true>(sv, op, o, d, inv, time, t_min, closest, pkey, depth, t); // (the op is the wave's)
        else hit = prim_hit_framed<false>(sv, op, po, pd, pinv, time, t_min, closest, t, sub);
        closest = hit ? t : closest;
        best = hit ? (op.leaf | (sub << 28)) : best;
    }
    // One XFORM step on the frame ray (fo, fd, finv), which the caller seeds with the world ray
    __device__ __forceinline__ void to_frame(const DevOp &op, D3 &fo, D3 &fd, D3 &finv) const {
        const DevXform x{op.d[0], op.d[1], op.d[2], op.d[3], op.d[4], op.d[5]};
        xform_ray<false>(x, fo, fd);
        if (x.s != 0.0) finv = d3(fast_rcp(fd.x), finv.y, fast_rcp(fd.z)); // a Y rotation leaves d.y alone
    }
    // A run of XFORM ops and the primitive behind it, as one step: x holds the run's first op, y receives every record
    // after it. On return x holds the record behind the primitive (loaded before the primitive's arithmetic was issued)
    // and kx its kind word. The frame ray lives in this step and nowhere else: nothing of it is carried round the
    // list loop, whose other steps test against the world ray.
    __device__ __forceinline__ void framed(const DevOp *ops, int32_t &c, DevOp &x, int &kx, DevOp &y) {
        const DevOp first = x; // the step's own copy: x is refilled at once with the record behind the primitive
        y = load_op_uniform(ops + c + 1);
        x = load_op_uniform(ops + c + 2); // an XFORM op is followed by a primitive and the sentinel at least
        D3 fo = o, fd = d, finv = inv;
        to_frame(first, fo, fd, finv);
        int ky = __builtin_amdgcn_readfirstlane(y.kind);
        c += 1;
        if ((ky & 0xff) == OP_XFORM) { // a chain of more than one wrapper: its records are fetched as they are met
            do {
                to_frame(y, fo, fd, finv);
                c += 1;
                y = load_op_uniform(ops + c);
                ky = __builtin_amdgcn_readfirstlane(y.kind);
            } while ((ky & 0xff) == OP_XFORM);
            x = load_op_uniform(ops + c + 1);
        }
        test(y, ky, fo, fd, finv);
        c += 1;
        kx = __builtin_amdgcn_readfirstlane(x.kind);
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
    ListWalk<MEDIA> w{sv, o, d, inv, dd, fast_rcp(dd), time, t_min, t_max, pkey, depth, -1};
    // The list ends with a sentinel EXIT op (a fused list holds no other), and the loop runs on the KIND of the
    // record it holds. Scalar loads return out of order, so the compiler can only ever wait for all of them: with
    // the loop test reading the current record, that wait falls before the loop and at the end of each step, after
    // the step's arithmetic has been issued, and never right behind the prefetch of the next record. Two records
    // alternate (a, b) so that no record is ever copied; a framed step consumes both and leaves the current record
    // where it was.
    DevOp a = load_op_uniform(ops), b;
    int ka = __builtin_amdgcn_readfirstlane(a.kind), kb;
    for (int32_t c = 0;;) {
        while ((ka & 0xff) == OP_XFORM) w.framed(ops, c, a, ka, b);
        if ((ka & 0xff) == OP_EXIT) break;
        b = load_op_uniform(ops + c + 1);
        w.test(a, ka, o, d, inv);
        c += 1;
        kb = __builtin_amdgcn_readfirstlane(b.kind);
        while ((kb & 0xff) == OP_XFORM) w.framed(ops, c, b, kb, a);
        if ((kb & 0xff) == OP_EXIT) break;
        a = load_op_uniform(ops + c + 1);
        w.test(b, kb, o, d, inv);
        c += 1;
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

// The slab tests of a 32-byte node (AccelNodeQ). A plane is gmin + q cell, so its distance along the ray is
// (gmin + q cell - o) inv = q (cell inv) + (gmin - o) inv: one FMA per plane with two per-ray constants per axis, A and B.
// 1 / d is clamped for them: along an axis the ray runs parallel to, an infinite factor would turn every product into
// NaN (0 inf, inf - inf), which min / max drop, and such a ray would then be culled by nothing on that axis (a ray with
// a zero direction by nothing at all); with the clamp the distances are +-huge with the right sign.
// They run in f32: the inner-node step of the kernels that read these nodes is bound by the vector ALU (an f64 min, max, FMA or
// conversion issues at half the rate of its f32 form, and there is no three-operand min / max in f64). They stay
// conservative by construction: a plane's distance q A + B (A = cell / d, B = (gmin - o) / d) evaluated in f32 from the
// rounded constants is off by less than E = (65536 |A| + |B|) 2^-21 on that axis (three roundings of 2^-24 relative to
// the larger of the two terms, with a factor of two to spare), so the near side of an axis is taken from planes
// evaluated with B - E and the far side from planes evaluated with B + E: the interval only grows, a box the f64 test
// accepts is accepted here, and since boxes only cull, every ray finds the hit it finds through the f64 test. The ray
// interval is widened the same way: t_min rounded down, the closest hit rounded up. 1 / d is clamped to +-1e25 (f32
// holds 65536 A and B up to scenes of 1e8 units; a parallel axis gives +-huge distances with the right sign).
struct QuantRayF {
    float a[3], bl[3], bh[3];
};
__device__ __forceinline__ QuantRayF quant_ray_f(const AccelGrid &g, D3 o, D3 inv) {
    const double big = 1.0e25;
    const double i[3] = {fmin(fmax(inv.x, -big), big), fmin(fmax(inv.y, -big), big), fmin(fmax(inv.z, -big), big)};
    const double og[3] = {g.gmin[0] - o.x, g.gmin[1] - o.y, g.gmin[2] - o.z};
    QuantRayF r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double A = g.cell[k] * i[k], B = og[k] * i[k];
        const double E = (65536.0 * fabs(A) + fabs(B)) * 4.76837158203125e-07; // 2^-21
        r.a[k] = (float)A;
        r.bl[k] = (float)(B - E);
        r.bh[k] = (float)(B + E);
    }
    return r;
}
// t_min_down, t_max_up: the ray interval (t_max = the closest hit so far), rounded outward.
// Which plane of an axis is the near one depends on the sign of the ray's direction alone, so the two planes are picked as
// whole words (both children at once) before they are converted: three shifts and six bit selects take the place of
// twelve more FMAs and twelve min / max.
// w[0..5]: the node's first six words as loaded (lo[x], lo[y], lo[z], hi[x], hi[y], hi[z], each holding child 0 in its low and
// child 1 in its high half); the node is read as two 16-byte loads by the caller (AccelNodeQWords).
struct AccelNodeQWords {
    uint4 a, b; // a = lo[0..2], hi[0]; b = hi[1..2], child[0..1]
};
static_assert(sizeof(AccelNodeQWords) == sizeof(AccelNodeQ), "two 16-byte words per node");
__device__ __forceinline__ void accel_boxes_qf(const AccelNodeQWords &n, const QuantRayF &r, float t_min_down, float t_max_up,
                                               bool &h0, bool &h1, float &t0, float &t1) {
    const uint32_t los[3] = {n.a.x, n.a.y, n.a.z}, his[3] = {n.a.w, n.b.x, n.b.y};
    float nr[2][3], fr[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t lo = los[k], hi = his[k];
        const uint32_t m = (uint32_t)((int32_t)__float_as_uint(r.a[k]) >> 31); // all ones: the ray runs down this axis
        const uint32_t nearw = (hi & m) | (lo & ~m), farw = (lo & m) | (hi & ~m);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float qn = (float)((nearw >> (16 * c)) & 0xffffu), qf = (float)((farw >> (16 * c)) & 0xffffu);
            nr[c][k] = __builtin_fmaf(qn, r.a[k], r.bl[k]);
            fr[c][k] = __builtin_fmaf(qf, r.a[k], r.bh[k]);
        }
    }
    t0 = fmaxf(fmaxf(nr[0][0], nr[0][1]), fmaxf(nr[0][2], t_min_down));
    t1 = fmaxf(fmaxf(nr[1][0], nr[1][1]), fmaxf(nr[1][2], t_min_down));
    const float f0 = fminf(fminf(fr[0][0], fr[0][1]), fminf(fr[0][2], t_max_up));
    const float f1 = fminf(fminf(fr[1][0], fr[1][1]), fminf(fr[1][2], t_max_up));
    h0 = t0 <= f0;
    h1 = t1 <= f1;
}
// an f64 distance as an f32 that is not below / not above it (beyond the f32 range: +-inf)
__device__ __forceinline__ float f32_not_below(double t) {
    const float f = (float)t;
    return __builtin_fmaf(fabsf(f), 1.1920929e-07f, f); // + 2^-23 |f|: a step or two up
}
__device__ __forceinline__ float f32_not_above(double t) {
    const float f = (float)t;
    return __builtin_fmaf(fabsf(f), -1.1920929e-07f, f);
}

// COUNT: node records read and primitives tested are added to a.counters (PT_CFG_COUNT_ACCEL). The lane pool form
// below walks the same tree in the same order ray by ray, so the counts describe both.
template <bool MEDIA, bool COUNT = false>
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
        uint32_t my_nodes = 0, my_prims = 0;
        // tests the primitives of a leaf child
        auto leaf = [&](int32_t first, int32_t count) {
            if constexpr (COUNT) my_prims += (uint32_t)count;
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
            if constexpr (COUNT) my_nodes += 1;
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
        if constexpr (COUNT) {
            atomicAdd(&a.counters->node_visits, (unsigned long long)my_nodes);
            atomicAdd(&a.counters->prim_tests, (unsigned long long)my_prims);
        }
    }
}

// The same traversal with the wave kept full. Rays of one wave need very different numbers of steps (a ray that
// leaves the scene at once next to one that grazes a thousand boxes), so in the form above most lanes of a wave sit
// idle for most of its life. Here a wave owns a run of the queue and works as a pool of 64 ray slots: a lane whose
// ray is done writes its result and, once enough lanes are free, the free lanes take the next rays of the run
// (ballot + prefix count, no atomics: the run belongs to this wave alone). Each round first steps every lane down
// through inner nodes until it stands on a leaf (or is done), then tests the leaves, so both loops run with most
// lanes active. Runs are dealt shard by shard; any grid is correct.
// QUANT: the nodes are read in their 32-byte form (`nodes_q`, `grid`); same hits.
template <bool MEDIA, bool QUANT = false>
__global__ __launch_bounds__(kBlock, MEDIA ? 4 : 5) void k_extend_accel_pool(Queue q, SceneView sv, const AccelNode *nodes,
                                                                            const DevOp *prims, ExtendArgs a,
                                                                            uint32_t tuning, const AccelNodeQ *nodes_q = nullptr,
                                                                            AccelGrid grid = AccelGrid{}) {
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
    QuantRayF qr{};
    float closest_up = 0.0f;                             // QUANT: the closest hit so far as an f32 that is not below it
    const float t_min_down = f32_not_above(a.t_min), t_max_up = f32_not_below(a.t_max);
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
                    if constexpr (QUANT) {
                        qr = quant_ray_f(grid, o, inv);
                        closest_up = t_max_up;
                    }
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
                bool h0, h1, first0;
                int32_t c0, c1;
                if constexpr (QUANT) {
                    const AccelNodeQWords n = ((const AccelNodeQWords *)nodes_q)[node];
                    float t0, t1;
                    accel_boxes_qf(n, qr, t_min_down, closest_up, h0, h1, t0, t1);
                    first0 = t0 <= t1;
                    c0 = (int32_t)n.b.z;
                    c1 = (int32_t)n.b.w;
                } else {
                    const AccelNode &n = nodes[node];
                    double t0, t1;
                    h0 = accel_box(n.lo[0], n.hi[0], o, inv, a.t_min, closest, t0);
                    h1 = accel_box(n.lo[1], n.hi[1], o, inv, a.t_min, closest, t1);
                    first0 = t0 <= t1;
                    c0 = n.child[0];
                    c1 = n.child[1];
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
                    if constexpr (QUANT) closest_up = f32_not_below(t);
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
