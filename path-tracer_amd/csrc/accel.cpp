// Host-side SAH BVH builder for the fast-mode traversal (see accel.h).
#include "accel.h"

#include <limits>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace ptdev {
namespace {

struct Box {
    double lo[3], hi[3];
    void clear() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = DBL_MAX;
            hi[k] = -DBL_MAX;
        }
    }
    void grow(const Box &b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    void grow_point(const double p[3]) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    double half_area() const {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// Bounds of a box after one chain step is undone (primitive space -> the enclosing space): the step maps
// p' = R (p - t) with R = [[c, 0, -s], [0, 1, 0], [s, 0, c]] (xform_ray), so p = R^T p' + t.
Box unxform_box(const DevXform &x, const Box &b) {
    Box out;
    out.clear();
    for (int corner = 0; corner < 8; ++corner) {
        const double px = (corner & 1) ? b.hi[0] : b.lo[0];
        const double py = (corner & 2) ? b.hi[1] : b.lo[1];
        const double pz = (corner & 4) ? b.hi[2] : b.lo[2];
        const double p[3] = {x.c * px + x.s * pz + x.tx, py + x.ty, -x.s * px + x.c * pz + x.tz};
        out.grow_point(p);
    }
    return out;
}

// Bounds of a primitive op in its own space.
bool local_bounds(const FlatScene &flat, const DevOp &op, Box &b) {
    const int kind = op_kind(op.kind);
    const double *d = op.d;
    switch (kind) {
    case OP_SPHERE: {
        const double r = std::fabs(d[3]);
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = d[k] - r;
            b.hi[k] = d[k] + r;
        }
        return true;
    }
    case OP_MSPHERE: {
        // center(t) = c0 + ((t - time0) / range) * e (src/sphere.rs:130-132), linear in t. Rays carry times of the
        // camera's shutter; the reference builds its boxes for (0, 1) (src/lib.rs:1031-1034 with bvh.rs:60-61), so
        // the box covers [min(time0, 0), max(time0 + range, 1)].
        const double *e = flat.leaves[op.leaf].e;
        const double r = std::fabs(d[3]);
        const double t_lo = std::min(d[4], 0.0), t_hi = std::max(d[4] + d[5], 1.0);
        b.clear();
        for (double t : {t_lo, t_hi}) {
            const double f = (t - d[4]) / d[5];
            for (int k = 0; k < 3; ++k) {
                const double c = d[k] + f * e[k];
                b.lo[k] = std::min(b.lo[k], c - r);
                b.hi[k] = std::max(b.hi[k], c + r);
            }
        }
        return true;
    }
    case OP_RECT: {
        const int axis = op_axis(op.kind); // plane axis; (a, b) axes as in src/rect.rs:71-85
        const int a_axis = axis == 0 ? 1 : 0, b_axis = axis == 2 ? 1 : 2;
        b.lo[a_axis] = d[0];
        b.hi[a_axis] = d[1];
        b.lo[b_axis] = d[2];
        b.hi[b_axis] = d[3];
        b.lo[axis] = d[4] - 0.0001; // the pad of src/rect.rs:116-122
        b.hi[axis] = d[4] + 0.0001;
        return true;
    }
    case OP_BOXPRIM:
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = d[k] - 0.0001; // the faces are rectangles: same pad
            b.hi[k] = d[k + 3] + 0.0001;
        }
        return true;
    case OP_MEDIUM: {
        // the boundary primitive, through the boundary's own chain
        if (op.skip < 0 || op.skip >= (int32_t)flat.ops.size()) return false;
        const DevOp &bop = flat.ops[op.skip];
        Box bb;
        if (!local_bounds(flat, bop, bb)) return false;
        const int cc = op_chain_count(bop.kind);
        for (int k = cc - 1; k >= 0; --k) bb = unxform_box(flat.xforms[(size_t)bop.chain_start + k], bb);
        b = bb;
        return true;
    }
    default:
        return false;
    }
}

struct BuildPrim {
    Box box;
    double centre[3];
    int32_t op; // index into flat.ops
};

// f32 values that bracket an f64 value
float round_down(double v) {
    float f = (float)v;
    return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double v) {
    float f = (float)v;
    return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}

struct Builder {
    std::vector<BuildPrim> prims;
    std::vector<AccelNode> nodes;
    std::vector<int32_t> order; // op indices in leaf order
    int max_depth = 0;
    bool too_deep = false;

    static constexpr int kBins = 16;
    static constexpr int kLeafMax = 4; // a leaf reference holds its count in 3 bits

    // Returns the child reference for prims[first, first + n): an inner node index or ~first_leaf_prim, plus count.
    void build_child(int first, int n, int depth, int32_t &child, int32_t &count, Box &box) {
        box.clear();
        Box cbox;
        cbox.clear();
        for (int i = first; i < first + n; ++i) {
            box.grow(prims[i].box);
            cbox.grow_point(prims[i].centre);
        }
        max_depth = std::max(max_depth, depth);
        int split = -1;
        // SAH splits may be lopsided; they are allowed only while a median-split subtree of these n primitives
        // (depth ceil(log2 n)) would still fit under the depth cap, which bounds the traversal stack.
        int log2n = 0;
        while ((1 << log2n) < n) ++log2n;
        if (n > 1 && depth < kAccelMaxDepth - 2) {
            if (depth + log2n + 2 < kAccelMaxDepth - 2) split = sah_split(first, n, box, cbox);
            if (split < 0 && n > kLeafMax) split = median_split(first, n, cbox);
        } else if (n > kLeafMax) {
            too_deep = true;
        }
        if (split < 0) {
            child = accel_leaf_ref((int32_t)order.size(), n);
            count = n;
            for (int i = first; i < first + n; ++i) order.push_back(prims[i].op);
            return;
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        std::memset(&nodes[me], 0, sizeof(AccelNode));
        int32_t c[2], k[2];
        Box b[2];
        build_child(first, split - first, depth + 1, c[0], k[0], b[0]);
        build_child(split, first + n - split, depth + 1, c[1], k[1], b[1]);
        AccelNode &nd = nodes[me];
        for (int s = 0; s < 2; ++s) {
            for (int a = 0; a < 3; ++a) {
                nd.lo[s][a] = round_down(b[s].lo[a]);
                nd.hi[s][a] = round_up(b[s].hi[a]);
            }
            nd.child[s] = c[s];
        }
        child = me;
        count = 0;
    }

    // Binned SAH over the three centroid axes; returns the partition point or -1 when a leaf is cheaper.
    int sah_split(int first, int n, const Box &box, const Box &cbox) {
        double best_cost = DBL_MAX;
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = cbox.lo[axis], extent = cbox.hi[axis] - cbox.lo[axis];
            if (!(extent > 0.0)) continue;
            Box bin_box[kBins];
            int bin_n[kBins];
            for (int b = 0; b < kBins; ++b) {
                bin_box[b].clear();
                bin_n[b] = 0;
            }
            const double scale = kBins / extent;
            for (int i = first; i < first + n; ++i) {
                int b = (int)((prims[i].centre[axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bin_box[b].grow(prims[i].box);
                bin_n[b]++;
            }
            double right_area[kBins];
            int right_n[kBins];
            Box acc;
            acc.clear();
            int cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bin_box[b]);
                cnt += bin_n[b];
                right_area[b] = cnt ? acc.half_area() : 0.0;
                right_n[b] = cnt;
            }
            acc.clear();
            cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bin_box[b]);
                cnt += bin_n[b];
                if (cnt == 0 || right_n[b + 1] == 0) continue;
                const double cost = acc.half_area() * cnt + right_area[b + 1] * right_n[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        if (best_axis < 0) return -1;
        // traversal step ~ one primitive test; keep small leaves when splitting does not pay
        const double leaf_cost = box.half_area() * n;
        if (n <= kLeafMax && best_cost + box.half_area() >= leaf_cost) return -1;
        const double lo = cbox.lo[best_axis], scale = kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
        auto mid = std::partition(prims.begin() + first, prims.begin() + first + n, [&](const BuildPrim &p) {
            int b = (int)((p.centre[best_axis] - lo) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            return b <= best_bin;
        });
        const int split = (int)(mid - prims.begin());
        if (split == first || split == first + n) return -1;
        return split;
    }

    int median_split(int first, int n, const Box &cbox) {
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
        const int mid = first + n / 2;
        std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + n,
                         [&](const BuildPrim &a, const BuildPrim &b) { return a.centre[axis] < b.centre[axis]; });
        return mid;
    }
};

} // namespace

bool accel_prim_bounds(const FlatScene &flat, const DevOp &op, double lo[3], double hi[3]) {
    if (op.leaf < 0 || op.leaf >= (int32_t)flat.leaves.size()) return false;
    Box b;
    if (!local_bounds(flat, op, b)) return false;
    const DevLeaf &lf = flat.leaves[op.leaf];
    for (int k = lf.full_count - 1; k >= 0; --k) b = unxform_box(flat.xforms[(size_t)lf.full_start + k], b);
    for (int k = 0; k < 3; ++k) {
        lo[k] = b.lo[k];
        hi[k] = b.hi[k];
    }
    return true;
}

// Parents and levels of a finished node array: where each node's own bounds are stored in its parent, and the nodes
// grouped by depth, deepest level first, for the bottom-up refit.
void accel_link_levels(Accel &out) {
    const int32_t nn = (int32_t)out.nodes.size();
    out.parent_slot.assign(nn, -1);
    std::vector<int32_t> level(nn, 0);
    std::vector<int32_t> todo{0};
    int32_t deepest = 0;
    while (!todo.empty()) {
        const int32_t id = todo.back();
        todo.pop_back();
        for (int s = 0; s < 2; ++s) {
            const int32_t c = out.nodes[id].child[s];
            if (c < 0) continue;
            out.parent_slot[c] = id * 2 + s;
            level[c] = level[id] + 1;
            deepest = std::max(deepest, level[c]);
            todo.push_back(c);
        }
    }
    out.level_start.assign((size_t)deepest + 2, 0);
    for (int32_t i = 0; i < nn; ++i) out.level_start[(size_t)(deepest - level[i]) + 1]++;
    for (size_t k = 1; k < out.level_start.size(); ++k) out.level_start[k] += out.level_start[k - 1];
    out.by_level.assign(nn, 0);
    std::vector<int32_t> fill(out.level_start.begin(), out.level_start.end() - 1);
    for (int32_t i = 0; i < nn; ++i) out.by_level[fill[(size_t)(deepest - level[i])]++] = i;
}

bool build_accel(const FlatScene &flat, Accel &out, std::string &error) {
    Builder bld;
    for (int32_t i = 0; i < flat.n_traversal_ops; ++i) {
        const DevOp &op = flat.ops[i];
        const int kind = op_kind(op.kind);
        if (kind == OP_BOX || kind == OP_ENTER || kind == OP_EXIT) continue;
        BuildPrim p;
        if (!accel_prim_bounds(flat, op, p.box.lo, p.box.hi)) {
            error = "primitive without bounds";
            return false;
        }
        for (int k = 0; k < 3; ++k) {
            if (!(p.box.lo[k] <= p.box.hi[k]) || !std::isfinite(p.box.lo[k]) || !std::isfinite(p.box.hi[k])) {
                error = "primitive with non-finite bounds";
                return false;
            }
            p.centre[k] = 0.5 * (p.box.lo[k] + p.box.hi[k]);
        }
        p.op = i;
        bld.prims.push_back(p);
    }
    if (bld.prims.size() < 2) {
        error = "fewer than two primitives";
        return false;
    }
    if (bld.prims.size() >= (size_t(1) << 28)) { // a leaf reference holds 28 bits of primitive index
        error = "too many primitives";
        return false;
    }
    int32_t child, count;
    Box box;
    // Primitives that span the scene (a fog medium around everything, the ground sphere of random_scene) would drag a
    // scene-sized box down every level the SAH needs to separate them from the primitives whose centres lie near
    // theirs: a spine of nodes every ray has to visit. They become the root's first child instead, one leaf, and the
    // tree over the other primitives its second. A ray is inside both boxes from the start, the traversal takes the
    // first child on a tie, so the lanes that start rays together also test these primitives together, and every
    // later box test has their hit distance to cull with.
    int n_wide = 0;
    {
        Box all;
        all.clear();
        for (const BuildPrim &q : bld.prims) all.grow(q.box);
        const double wide = 0.5 * all.half_area();
        auto mid = std::stable_partition(bld.prims.begin(), bld.prims.end(),
                                         [&](const BuildPrim &q) { return q.box.half_area() >= wide; });
        n_wide = (int)(mid - bld.prims.begin());
    }
    const int n_all = (int)bld.prims.size();
    if (n_wide >= 1 && n_wide <= Builder::kLeafMax && n_all - n_wide >= 2) {
        bld.nodes.emplace_back();
        std::memset(&bld.nodes[0], 0, sizeof(AccelNode));
        Box b[2];
        int32_t c[2];
        b[0].clear();
        c[0] = accel_leaf_ref(0, n_wide);
        for (int i = 0; i < n_wide; ++i) {
            b[0].grow(bld.prims[i].box);
            bld.order.push_back(bld.prims[i].op);
        }
        bld.build_child(n_wide, n_all - n_wide, 1, c[1], count, b[1]);
        AccelNode &root = bld.nodes[0];
        for (int s = 0; s < 2; ++s) {
            for (int a = 0; a < 3; ++a) {
                root.lo[s][a] = round_down(b[s].lo[a]);
                root.hi[s][a] = round_up(b[s].hi[a]);
            }
            root.child[s] = c[s];
        }
        bld.max_depth = std::max(bld.max_depth, 1);
        child = 0;
    } else {
        bld.build_child(0, n_all, 0, child, count, box);
    }
    if (bld.too_deep || child != 0) { // child != 0: the whole scene became one leaf
        error = "tree too deep or degenerate";
        return false;
    }
    out.nodes.swap(bld.nodes);
    out.depth = bld.max_depth;
    accel_link_levels(out);
    out.prims.clear();
    out.prims.reserve(bld.order.size());
    for (int32_t i : bld.order) {
        DevOp op = flat.ops[i];
        const DevLeaf &lf = flat.leaves[op.leaf];
        // self-contained: the op applies the full world -> primitive chain itself (no ENTER / EXIT frames here)
        op.kind = op_pack(op_kind(op.kind), op_axis(op.kind), lf.full_count);
        op.chain_start = lf.full_start;
        if (op_kind(op.kind) == OP_SPHERE && lf.full_count > 0) {
            // A sphere under translations and rotations is a sphere: its centre is carried out to world space once,
            // here, and the traversal tests it against the world ray with no chain at all (the hit distance along
            // the ray is the same in both frames; the hit record is still built in the primitive's frame).
            double c[3] = {op.d[0], op.d[1], op.d[2]};
            for (int k = lf.full_count - 1; k >= 0; --k) {
                const DevXform &x = flat.xforms[(size_t)lf.full_start + k];
                const double wx = x.c * c[0] + x.s * c[2], wz = -x.s * c[0] + x.c * c[2]; // hitable.rs:189-194
                c[0] = wx + x.tx;
                c[1] = c[1] + x.ty;
                c[2] = wz + x.tz;
            }
            op.d[0] = c[0];
            op.d[1] = c[1];
            op.d[2] = c[2];
            op.kind = op_pack(OP_SPHERE, 0, 0);
            op.chain_start = 0;
        }
        out.prims.push_back(op);
    }
    // Compact records, if the scene is plain spheres alone: untransformed, one of the four surface materials, sign bits
    // free to carry the material kind, and a colour that is either constant or a texture evaluated from the hit point
    // alone (a checker: random_scene's ground). A textured sphere's record holds the texture's index in albedo[0] and a
    // NaN in albedo[1] (a constant colour that qualifies is finite).
    out.spheres.clear();
    bool compact = !flat.has_medium;
    for (size_t k = 0; k < out.prims.size() && compact; ++k) {
        const DevOp &op = out.prims[k];
        const DevLeaf &lf = flat.leaves[op.leaf];
        const double param = lf.mat_kind == PT_M_LAMBERTIAN ? lf.emissive
                             : lf.mat_kind == PT_M_METAL    ? lf.fuzz
                             : lf.mat_kind == PT_M_DIELECTRIC ? lf.ref_idx : 0.0;
        const bool textured = lf.texture >= 0;
        compact = op_kind(op.kind) == OP_SPHERE && lf.full_count == 0 && lf.mat_kind >= PT_M_LAMBERTIAN &&
                  lf.mat_kind <= PT_M_DIELECTRIC && std::isfinite(op.d[3]) && !std::signbit(op.d[3]) && std::isfinite(param) &&
                  !std::signbit(param) &&
                  (textured ? (lf.tex_flags & (kLeafTexUv | kLeafTexNoise)) == 0
                            : std::isfinite(lf.albedo[0]) && std::isfinite(lf.albedo[1]) && std::isfinite(lf.albedo[2]));
        if (!compact) break;
        const int code = lf.mat_kind - 1;
        DevSphere sp;
        for (int a = 0; a < 3; ++a) {
            sp.c[a] = op.d[a];
            sp.albedo[a] = lf.albedo[a];
        }
        if (textured) {
            sp.albedo[0] = (double)lf.texture;
            sp.albedo[1] = std::numeric_limits<double>::quiet_NaN();
            sp.albedo[2] = 0.0;
        }
        sp.radius = (code & 1) ? -op.d[3] : op.d[3];
        sp.param = (code & 2) ? -param : param;
        out.spheres.push_back(sp);
    }
    if (!compact) out.spheres.clear();
    return true;
}

} // namespace ptdev

// Builds the tree for a scene and verifies its invariants on the host (no GPU): every primitive op sits in exactly
// one leaf, every child box contains the bounds of everything below it, and the depth respects the stack cap.
extern "C" int pt_scene_accel_info(const pt_scene_desc *scene, int32_t *n_nodes, int32_t *n_prims, int32_t *depth) {
    using namespace ptdev;
    FlatScene flat;
    std::string err;
    int rc = flatten_scene(scene, flat, err);
    if (rc != PT_OK) return rc;
    Accel acc;
    if (!build_accel(flat, acc, err)) return PT_ERR_UNSUPPORTED_SCENE;
    std::vector<int> seen(flat.leaves.size(), 0);
    struct Item { int32_t node; int depth; };
    std::vector<Item> todo{{0, 1}};
    int max_depth = 0;
    // bounds of a subtree, recomputed bottom-up with an explicit post-order walk
    std::vector<Box> sub(acc.nodes.size());
    std::vector<int> state(acc.nodes.size(), 0);
    std::vector<int32_t> walk{0};
    auto leaf_box = [&](int32_t first, int32_t count, Box &b) -> bool {
        b.clear();
        for (int32_t k = first; k < first + count; ++k) {
            if (k < 0 || k >= (int32_t)acc.prims.size()) return false;
            const DevOp &op = acc.prims[k];
            if (op.leaf < 0 || op.leaf >= (int32_t)seen.size()) return false;
            seen[op.leaf]++;
            Box pb;
            if (!accel_prim_bounds(flat, flat.ops[flat.leaves[op.leaf].op], pb.lo, pb.hi)) return false;
            b.grow(pb);
        }
        return true;
    };
    auto inside = [](const Box &inner, const double *lo, const double *hi) {
        for (int a = 0; a < 3; ++a)
            if (inner.lo[a] < lo[a] || inner.hi[a] > hi[a]) return false;
        return true;
    };
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        max_depth = std::max(max_depth, it.depth);
        if (it.node < 0 || it.node >= (int32_t)acc.nodes.size()) return PT_ERR_INVALID_ARG;
        const AccelNode &n = acc.nodes[it.node];
        for (int s = 0; s < 2; ++s)
            if (n.child[s] >= 0) todo.push_back({n.child[s], it.depth + 1});
    }
    while (!walk.empty()) {
        const int32_t id = walk.back();
        const AccelNode &n = acc.nodes[id];
        if (state[id] == 0) {
            state[id] = 1;
            for (int s = 0; s < 2; ++s)
                if (n.child[s] >= 0) walk.push_back(n.child[s]);
            continue;
        }
        walk.pop_back();
        if (state[id] == 2) continue;
        state[id] = 2;
        sub[id].clear();
        for (int s = 0; s < 2; ++s) {
            Box b;
            if (n.child[s] >= 0) b = sub[n.child[s]];
            else if (!leaf_box((~n.child[s]) >> 3, (~n.child[s]) & 7, b)) return PT_ERR_INVALID_ARG;
            const double lo[3] = {n.lo[s][0], n.lo[s][1], n.lo[s][2]}, hi[3] = {n.hi[s][0], n.hi[s][1], n.hi[s][2]};
            if (!inside(b, lo, hi)) return PT_ERR_INVALID_ARG;
            sub[id].grow(b);
        }
    }
    size_t expected = 0;
    for (int32_t i = 0; i < flat.n_traversal_ops; ++i) {
        const int kind = op_kind(flat.ops[i].kind);
        if (kind == OP_BOX || kind == OP_ENTER || kind == OP_EXIT) continue;
        ++expected;
        if (seen[flat.ops[i].leaf] != 1) return PT_ERR_INVALID_ARG;
    }
    if (expected != acc.prims.size() || max_depth >= kAccelMaxDepth) return PT_ERR_INVALID_ARG;
    if (n_nodes) *n_nodes = (int32_t)acc.nodes.size();
    if (n_prims) *n_prims = (int32_t)acc.prims.size();
    if (depth) *depth = max_depth;
    return PT_OK;
}
