// Fast-mode tree maintenance on the device: bottom-up refit and the Morton-order build.
// Part of ptgpu.hip (included inside its unnamed namespace, in this order); not a standalone header.

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

// The 32-byte form of every node (accel.h), made from the f32 nodes wherever those came from (host build, refit,
// device build). One step of slack on each side covers the rounding of the decode (gmin + steps * cell in f64).
__global__ __launch_bounds__(kBlock) void k_accel_quantise(const AccelNode *nodes, AccelNodeQ *out, int32_t n, AccelGrid g) {
    for (int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x); i < n; i += (int32_t)(gridDim.x * kBlock)) {
        const AccelNode nd = nodes[i];
        AccelNodeQ q;
        for (int s = 0; s < 2; ++s) {
            for (int a = 0; a < 3; ++a) {
                const double lo = floor(((double)nd.lo[s][a] - g.gmin[a]) / g.cell[a]) - 1.0;
                const double hi = ceil(((double)nd.hi[s][a] - g.gmin[a]) / g.cell[a]) + 1.0;
                // (an empty child box, lo = +max and hi = -max, becomes [65535, 0]: still empty)
                q.lo[a][s] = (uint16_t)fmin(fmax(lo, 0.0), 65535.0);
                q.hi[a][s] = (uint16_t)fmin(fmax(hi, 0.0), 65535.0);
            }
            q.child[s] = nd.child[s];
        }
        out[i] = q;
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
