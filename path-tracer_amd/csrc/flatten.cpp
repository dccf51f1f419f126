// Scene graph -> threaded op stream. See device_scene.h for the layout and its rationale.
#include "flatten.h"

#include <cmath>
#include <cstring>

namespace ptdev {

namespace {

// A wrapper as met in the graph; chains of these are compressed into DevXform steps when stored.
enum SimpleKind { S_TRANSLATE, S_ROTATE_Y, S_FLIP };
struct SimpleOp {
    int kind;
    double a, b, c;
};

DevXform identity_step() { return DevXform{0.0, 0.0, 0.0, 0.0, 1.0, 1.0}; }

// [Translate][RotateY] in that order share a step; anything else starts a new one. Flips only negate the
// final normal, so their parity rides on the last step (or on an identity step if the chain has no other).
std::vector<DevXform> compress_chain(const std::vector<SimpleOp> &ops) {
    std::vector<DevXform> out;
    bool open = false, has_t = false, has_r = false;
    double sign = 1.0;
    DevXform cur = identity_step();
    auto flush = [&]() {
        if (open) out.push_back(cur);
        cur = identity_step();
        open = has_t = has_r = false;
    };
    for (const SimpleOp &o : ops) {
        if (o.kind == S_FLIP) {
            sign = -sign;
        } else if (o.kind == S_TRANSLATE) {
            if (has_t || has_r) flush();
            cur.tx = o.a; cur.ty = o.b; cur.tz = o.c;
            open = has_t = true;
        } else {
            if (has_r) flush();
            cur.s = o.a; cur.c = o.b;
            open = has_r = true;
        }
    }
    flush();
    if (sign < 0.0) {
        if (out.empty()) out.push_back(identity_step());
        out.back().sign = -1.0;
    }
    return out;
}

struct FrameRef {
    int32_t start = 0; // range of the frame's full chain in xforms[]
    int32_t count = 0;
};

class Flattener {
  public:
    Flattener(const pt_scene_desc *s, FlatScene &out, std::string &err) : s_(s), out_(out), err_(err) {}

    int run() {
        if (!s_ || s_->struct_size != sizeof(pt_scene_desc)) return fail(PT_ERR_INVALID_ARG, "bad pt_scene_desc size");
        if (s_->n_hitables <= 0 || !s_->hitables) return fail(PT_ERR_INVALID_ARG, "scene has no hitables");
        if (!node_ok(s_->root)) return fail(PT_ERR_INVALID_ARG, "scene root out of range");
        if (s_->n_materials < 0 || s_->n_textures < 0 || s_->n_lights < 0)
            return fail(PT_ERR_INVALID_ARG, "negative table size");
        for (int i = 0; i < s_->n_materials; ++i) {
            const pt_material &m = s_->materials[i];
            bool needs_tex = m.kind == PT_M_LAMBERTIAN || m.kind == PT_M_DIFFUSE_LIGHT || m.kind == PT_M_ISOTROPIC;
            if (m.kind < PT_M_DUMMY || m.kind > PT_M_ISOTROPIC) return fail(PT_ERR_INVALID_ARG, "bad material kind");
            if (needs_tex && (m.texture < 0 || m.texture >= s_->n_textures))
                return fail(PT_ERR_INVALID_ARG, "material texture out of range");
        }
        for (int i = 0; i < s_->n_textures; ++i) {
            const pt_texture &t = s_->textures[i];
            if (t.kind < PT_T_CONSTANT || t.kind > PT_T_IMAGE) return fail(PT_ERR_INVALID_ARG, "bad texture kind");
            if (t.kind == PT_T_CHECKER &&
                (t.even < 0 || t.even >= s_->n_textures || t.odd < 0 || t.odd >= s_->n_textures))
                return fail(PT_ERR_INVALID_ARG, "checker child texture out of range");
            if (t.kind == PT_T_CHECKER && (t.even >= i || t.odd >= i))
                return fail(PT_ERR_UNSUPPORTED_SCENE, "checker children must be created before the checker");
            if (t.kind == PT_T_NOISE && !s_->perlin) return fail(PT_ERR_INVALID_ARG, "noise texture without perlin tables");
            if (t.kind == PT_T_IMAGE) {
                if (t.width <= 0 || t.height <= 0 || !s_->texels ||
                    t.texel_offset + (uint64_t)t.width * t.height * 3 > s_->n_texel_bytes)
                    return fail(PT_ERR_INVALID_ARG, "image texture outside the texel blob");
            }
        }
        std::vector<SimpleOp> local;
        std::vector<DevXform> full;
        int rc = emit(s_->root, local, full, FrameRef{}, 0, true, 0);
        if (rc != PT_OK) return rc;
        out_.n_traversal_ops = (int32_t)out_.ops.size();
        // boundary ops of media are stored behind the traversal stream
        for (auto &op : out_.ops)
            if (op_kind(op.kind) == OP_MEDIUM) op.skip += out_.n_traversal_ops;
        for (auto &op : boundary_ops_) // the BOX ops of a boundary's sub-stream jump inside it
            if (op_kind(op.kind) == OP_BOX) op.skip += out_.n_traversal_ops;
        out_.ops.insert(out_.ops.end(), boundary_ops_.begin(), boundary_ops_.end());
        if (out_.ops.empty()) return fail(PT_ERR_INVALID_ARG, "scene flattened to nothing");

        for (int i = 0; i < s_->n_lights; ++i) {
            int h = s_->lights[i];
            if (!node_ok(h)) return fail(PT_ERR_INVALID_ARG, "light node out of range");
            const pt_hitable &n = s_->hitables[h];
            DevLight l;
            std::memset(&l, 0, sizeof(l));
            if (n.kind == PT_H_SPHERE) {
                l.kind = LIGHT_SPHERE;
                for (int k = 0; k < 4; ++k) l.d[k] = n.f[k];
            } else if (n.kind == PT_H_RECT) {
                l.kind = LIGHT_RECT;
                l.axis = n.axis;
                for (int k = 0; k < 5; ++k) l.d[k] = n.f[k];
            } else if (n.kind == PT_H_LIST) {
                return fail(PT_ERR_UNSUPPORTED_SCENE, "nested HitableList in the light list");
            } else {
                l.kind = LIGHT_OTHER; // trait defaults: pdf_value 0, random (1,0,0) (src/hitable.rs:30-35)
            }
            out_.lights.push_back(l);
        }
        return PT_OK;
    }

  private:
    bool node_ok(int h) const { return h >= 0 && h < s_->n_hitables; }
    int fail(int code, const std::string &msg) {
        if (err_.empty()) err_ = msg;
        return code;
    }

    // Appends `full` ++ `local` to xforms[] and returns where it starts (0 if the chain is empty).
    int32_t store_chain(const std::vector<DevXform> &full, const std::vector<DevXform> &local) { // both compressed
        if (full.empty() && local.empty()) return 0;
        int32_t start = (int32_t)out_.xforms.size();
        out_.xforms.insert(out_.xforms.end(), full.begin(), full.end());
        out_.xforms.insert(out_.xforms.end(), local.begin(), local.end());
        return start;
    }

    static SimpleOp xf(int kind, double a, double b, double c) { return SimpleOp{kind, a, b, c}; }

    // Fills geometry parameters of a primitive op from a graph node. Returns false if not a primitive.
    bool geometry_op(int h, DevOp &op, DevLeaf &leaf, int &rc) {
        const pt_hitable &n = s_->hitables[h];
        std::memset(&op, 0, sizeof(op));
        std::memset(&leaf, 0, sizeof(leaf));
        op.skip = -1;
        for (int k = 0; k < 6; ++k) leaf.prim[k] = h;
        leaf.material = n.material;
        rc = PT_OK;
        if ((n.kind == PT_H_SPHERE || n.kind == PT_H_MOVING_SPHERE || n.kind == PT_H_RECT || n.kind == PT_H_BOX) &&
            (n.material < 0 || n.material >= s_->n_materials)) {
            rc = fail(PT_ERR_INVALID_ARG, "primitive material out of range");
            return true;
        }
        switch (n.kind) {
        case PT_H_SPHERE:
            leaf.kind = OP_SPHERE;
            for (int k = 0; k < 4; ++k) op.d[k] = n.f[k];
            return true;
        case PT_H_MOVING_SPHERE:
            leaf.kind = OP_MSPHERE;
            op.d[0] = n.f[0]; op.d[1] = n.f[1]; op.d[2] = n.f[2]; op.d[3] = n.f[3];
            op.d[4] = n.f[7]; op.d[5] = n.f[8];
            leaf.e[0] = n.f[4]; leaf.e[1] = n.f[5]; leaf.e[2] = n.f[6];
            return true;
        case PT_H_RECT:
            if (n.axis < 0 || n.axis > 2) {
                rc = fail(PT_ERR_INVALID_ARG, "rect axis out of range");
                return true;
            }
            leaf.kind = OP_RECT;
            leaf.axis = n.axis;
            for (int k = 0; k < 5; ++k) op.d[k] = n.f[k];
            return true;
        case PT_H_BOX: {
            leaf.kind = OP_BOXPRIM;
            for (int k = 0; k < 6; ++k) op.d[k] = n.f[k];
            rc = check_box(h, leaf);
            return true;
        }
        default:
            return false;
        }
    }

    // The device intersects a box as the six rectangles AxisAlignedBox::new builds (src/axis_aligned_box.rs:17-22);
    // verify the graph's face list is exactly that, and record each face's node for pt_hit.prim.
    int check_box(int h, DevLeaf &leaf) {
        const pt_hitable &n = s_->hitables[h];
        if (n.list_count != 6 || n.list_start < 0 || n.list_start + 6 > s_->n_child_index)
            return fail(PT_ERR_UNSUPPORTED_SCENE, "box without its six faces");
        const double *lo = &n.f[0], *hi = &n.f[3];
        // face k: {a axis, b axis, plane axis, plane at hi?, flipped}
        const int spec[6][5] = {{0, 1, 2, 1, 0}, {0, 1, 2, 0, 1}, {0, 2, 1, 1, 0},
                                {0, 2, 1, 0, 1}, {1, 2, 0, 1, 0}, {1, 2, 0, 0, 1}};
        for (int k = 0; k < 6; ++k) {
            int c = s_->child_index[n.list_start + k];
            if (!node_ok(c)) return fail(PT_ERR_INVALID_ARG, "box face out of range");
            bool flipped = false;
            if (s_->hitables[c].kind == PT_H_FLIP) {
                flipped = true;
                c = s_->hitables[c].child[0];
                if (!node_ok(c)) return fail(PT_ERR_INVALID_ARG, "box face out of range");
            }
            const pt_hitable &r = s_->hitables[c];
            const int *sp = spec[k];
            bool ok = r.kind == PT_H_RECT && r.axis == sp[2] && flipped == (sp[4] == 1) && r.f[0] == lo[sp[0]] &&
                      r.f[1] == hi[sp[0]] && r.f[2] == lo[sp[1]] && r.f[3] == hi[sp[1]] &&
                      r.f[4] == (sp[3] ? hi[sp[2]] : lo[sp[2]]) && r.material == n.material;
            if (!ok) return fail(PT_ERR_UNSUPPORTED_SCENE, "box faces differ from the canonical six rectangles");
            leaf.prim[k] = c;
        }
        return PT_OK;
    }

    int emit(int h, std::vector<SimpleOp> local, const std::vector<DevXform> &full, FrameRef frame, int frame_depth,
             bool is_root, int guard) {
        if (!node_ok(h)) return fail(PT_ERR_INVALID_ARG, "hitable index out of range");
        if (guard > 4096) return fail(PT_ERR_INVALID_ARG, "hitable graph too deep or cyclic");
        const pt_hitable &n = s_->hitables[h];
        switch (n.kind) {
        case PT_H_FLIP:
            local.push_back(xf(S_FLIP, 0, 0, 0));
            return emit(n.child[0], local, full, frame, frame_depth, false, guard + 1);
        case PT_H_TRANSLATE:
            local.push_back(xf(S_TRANSLATE, n.f[0], n.f[1], n.f[2]));
            return emit(n.child[0], local, full, frame, frame_depth, false, guard + 1);
        case PT_H_ROTATE_Y:
            local.push_back(xf(S_ROTATE_Y, n.f[0], n.f[1], 0.0));
            return emit(n.child[0], local, full, frame, frame_depth, false, guard + 1);
        case PT_H_BVH: {
            if (!local.empty()) {
                // wrapped subtree: ENTER (apply the wrappers to the ray), the subtree, EXIT (restore)
                const std::vector<DevXform> steps = compress_chain(local);
                if (full.size() + steps.size() > 255)
                    return fail(PT_ERR_UNSUPPORTED_SCENE, "transform chain longer than 255");
                int32_t start = store_chain(full, steps);
                std::vector<DevXform> new_full(full);
                new_full.insert(new_full.end(), steps.begin(), steps.end());
                DevOp enter;
                std::memset(&enter, 0, sizeof(enter));
                enter.kind = op_pack(OP_ENTER, 0, (int)steps.size());
                enter.skip = -1;
                enter.leaf = -1;
                enter.chain_start = start + (int32_t)full.size();
                out_.ops.push_back(enter);
                out_.has_frames = true;
                if (frame_depth + 1 > out_.max_frame_depth) out_.max_frame_depth = frame_depth + 1;
                FrameRef inner{start, (int32_t)new_full.size()};
                int rc = emit(h, std::vector<SimpleOp>(), new_full, inner, frame_depth + 1, false, guard + 1);
                if (rc != PT_OK) return rc;
                DevOp exit_op;
                std::memset(&exit_op, 0, sizeof(exit_op));
                exit_op.kind = op_pack(OP_EXIT, 0, frame.count);
                exit_op.skip = -1;
                exit_op.leaf = -1;
                exit_op.chain_start = frame.start;
                out_.ops.push_back(exit_op);
                return PT_OK;
            }
            DevOp box;
            std::memset(&box, 0, sizeof(box));
            box.kind = op_pack(OP_BOX, 0, 0);
            box.leaf = -1;
            for (int k = 0; k < 6; ++k) box.d[k] = n.f[k];
            size_t at = out_.ops.size();
            out_.ops.push_back(box);
            int rc = emit(n.child[0], std::vector<SimpleOp>(), full, frame, frame_depth, false, guard + 1);
            if (rc != PT_OK) return rc;
            if (n.child[1] != n.child[0]) { // bvh.rs:47-49 duplicates a lone leaf; testing it once is equivalent
                rc = emit(n.child[1], std::vector<SimpleOp>(), full, frame, frame_depth, false, guard + 1);
                if (rc != PT_OK) return rc;
            }
            if (out_.ops.size() > (size_t)INT32_MAX) return fail(PT_ERR_TOO_LARGE, "too many ops");
            out_.ops[at].skip = (int32_t)out_.ops.size();
            return PT_OK;
        }
        case PT_H_LIST: {
            // HitableList::hit (src/hitable.rs:51-62) tests its members in order with a shrinking t_max, which is what a
            // run of ops in the stream does: a list, wherever it stands, is its members written out. Wrappers above it
            // distribute over the members: each wrapper transforms the ray once and the winning member's record back,
            // the same operations on the same values as with the wrapper around every member (t is untouched by them).
            if (n.list_start < 0 || n.list_count < 0 || n.list_start + n.list_count > s_->n_child_index)
                return fail(PT_ERR_INVALID_ARG, "list range out of bounds");
            for (int k = 0; k < n.list_count; ++k) {
                int rc = emit(s_->child_index[n.list_start + k], local, full, frame, frame_depth, false, guard + 1);
                if (rc != PT_OK) return rc;
            }
            return PT_OK;
        }
        case PT_H_MEDIUM: {
            if (n.material < 0 || n.material >= s_->n_materials)
                return fail(PT_ERR_INVALID_ARG, "medium phase function out of range");
            const std::vector<DevXform> steps = compress_chain(local);
            if (full.size() + steps.size() > 255) return fail(PT_ERR_UNSUPPORTED_SCENE, "transform chain longer than 255");
            // boundary: any subtree of wrappers, lists, trees and primitives, as a sub-stream in the medium's own frame
            // (ConstantMedium::hit walks it twice, src/volume.rs:30-31); one (wrapped) primitive in every recipe
            const int32_t bindex = (int32_t)boundary_ops_.size();
            int rc = emit_boundary(n.child[0], std::vector<SimpleOp>(), 0);
            if (rc != PT_OK) return rc;
            const int32_t bcount = (int32_t)boundary_ops_.size() - bindex;
            if (bcount < 1) return fail(PT_ERR_UNSUPPORTED_SCENE, "medium boundary without a primitive");
            if (bcount > 1) out_.general_boundaries = true;
            int32_t start = store_chain(full, steps);
            DevLeaf leaf;
            std::memset(&leaf, 0, sizeof(leaf));
            leaf.kind = OP_MEDIUM;
            leaf.material = n.material;
            leaf.full_start = start;
            leaf.full_count = (int32_t)(full.size() + steps.size());
            for (int k = 0; k < 6; ++k) leaf.prim[k] = h;
            leaf.op = (int32_t)out_.ops.size();
            DevOp op;
            std::memset(&op, 0, sizeof(op));
            op.kind = op_pack(OP_MEDIUM, 0, (int)steps.size());
            op.d[0] = n.f[0];
            op.d[1] = (double)bcount;
            op.skip = bindex; // relocated once the traversal stream is complete
            op.leaf = (int32_t)out_.leaves.size();
            op.chain_start = start + (int32_t)full.size();
            out_.leaves.push_back(leaf);
            out_.ops.push_back(op);
            out_.has_medium = true;
            return PT_OK;
        }
        default: {
            DevOp op;
            DevLeaf leaf;
            int rc = PT_OK;
            if (!geometry_op(h, op, leaf, rc)) return fail(PT_ERR_INVALID_ARG, "unknown hitable kind");
            if (rc != PT_OK) return rc;
            const std::vector<DevXform> steps = compress_chain(local);
            if (full.size() + steps.size() > 255) return fail(PT_ERR_UNSUPPORTED_SCENE, "transform chain longer than 255");
            if (out_.leaves.size() >= (size_t)(1 << 28)) return fail(PT_ERR_TOO_LARGE, "too many primitives");
            int32_t start = store_chain(full, steps);
            leaf.full_start = start;
            leaf.full_count = (int32_t)(full.size() + steps.size());
            leaf.op = (int32_t)out_.ops.size();
            op.kind = op_pack(leaf.kind, leaf.axis, (int)steps.size());
            op.chain_start = start + (int32_t)full.size();
            op.leaf = (int32_t)out_.leaves.size();
            out_.leaves.push_back(leaf);
            out_.ops.push_back(op);
            return PT_OK;
        }
        }
    }

    // A medium's boundary, into boundary_ops_: primitives carry their wrapper chains (relative to the medium's frame),
    // a list is its members in order, a tree its BOX ops with skips inside the sub-stream.
    int emit_boundary(int h, std::vector<SimpleOp> local, int guard) {
        if (!node_ok(h)) return fail(PT_ERR_INVALID_ARG, "medium boundary out of range");
        if (guard > 4096) return fail(PT_ERR_INVALID_ARG, "hitable graph too deep or cyclic");
        const pt_hitable &n = s_->hitables[h];
        switch (n.kind) {
        case PT_H_FLIP:
            local.push_back(xf(S_FLIP, 0, 0, 0));
            return emit_boundary(n.child[0], local, guard + 1);
        case PT_H_TRANSLATE:
            local.push_back(xf(S_TRANSLATE, n.f[0], n.f[1], n.f[2]));
            return emit_boundary(n.child[0], local, guard + 1);
        case PT_H_ROTATE_Y:
            local.push_back(xf(S_ROTATE_Y, n.f[0], n.f[1], 0.0));
            return emit_boundary(n.child[0], local, guard + 1);
        case PT_H_LIST:
            if (n.list_start < 0 || n.list_count < 0 || n.list_start + n.list_count > s_->n_child_index)
                return fail(PT_ERR_INVALID_ARG, "list range out of bounds");
            for (int k = 0; k < n.list_count; ++k)
                if (int rc = emit_boundary(s_->child_index[n.list_start + k], local, guard + 1); rc != PT_OK) return rc;
            return PT_OK;
        case PT_H_BVH: {
            // A tree under wrappers (Translate / RotateY / FlipNormals above a BvhNode): its BOX ops carry the wrappers as a
            // chain of their own, applied to the ray before the slab test, and its members inherit them in front of theirs:
            // the reference tests every box and primitive of the subtree with the ray the wrappers hand down
            // (src/hitable.rs:98-117,151-181, src/bvh.rs:74-99).
            const std::vector<DevXform> bsteps = compress_chain(local);
            if (bsteps.size() > 255) return fail(PT_ERR_UNSUPPORTED_SCENE, "transform chain longer than 255");
            DevOp box;
            std::memset(&box, 0, sizeof(box));
            box.kind = op_pack(OP_BOX, 0, (int)bsteps.size());
            box.leaf = -1;
            box.chain_start = store_chain(std::vector<DevXform>(), bsteps);
            for (int k = 0; k < 6; ++k) box.d[k] = n.f[k];
            const size_t at = boundary_ops_.size();
            boundary_ops_.push_back(box);
            if (int rc = emit_boundary(n.child[0], local, guard + 1); rc != PT_OK) return rc;
            if (n.child[1] != n.child[0])
                if (int rc = emit_boundary(n.child[1], local, guard + 1); rc != PT_OK) return rc;
            boundary_ops_[at].skip = (int32_t)boundary_ops_.size(); // relocated with the rest of the sub-stream
            return PT_OK;
        }
        case PT_H_MEDIUM:
            return fail(PT_ERR_UNSUPPORTED_SCENE, "a medium as (part of) a medium's boundary");
        default: {
            DevOp bop;
            DevLeaf bleaf;
            int rc = PT_OK;
            if (!geometry_op(h, bop, bleaf, rc)) return fail(PT_ERR_INVALID_ARG, "unknown hitable kind");
            if (rc != PT_OK) return rc;
            const std::vector<DevXform> isteps = compress_chain(local);
            if (isteps.size() > 255) return fail(PT_ERR_UNSUPPORTED_SCENE, "transform chain longer than 255");
            if (out_.leaves.size() >= (size_t)(1 << 28)) return fail(PT_ERR_TOO_LARGE, "too many primitives");
            const int32_t istart = store_chain(std::vector<DevXform>(), isteps);
            bleaf.op = -1; // set once the boundary ops have their final places
            bleaf.full_start = istart;
            bleaf.full_count = (int32_t)isteps.size();
            bop.kind = op_pack(bleaf.kind, bleaf.axis, (int)isteps.size());
            bop.chain_start = istart;
            bop.leaf = (int32_t)out_.leaves.size();
            out_.leaves.push_back(bleaf);
            boundary_ops_.push_back(bop);
            return PT_OK;
        }
        }
    }

    const pt_scene_desc *s_;
    FlatScene &out_;
    std::string &err_;
    std::vector<DevOp> boundary_ops_;
};

} // namespace

int flatten_scene(const pt_scene_desc *scene, FlatScene &out, std::string &error) {
    out = FlatScene();
    Flattener f(scene, out, error);
    int rc = f.run();
    if (rc != PT_OK) return rc;
    // boundary leaves point at their op once the boundary ops have been appended
    for (size_t i = (size_t)out.n_traversal_ops; i < out.ops.size(); ++i)
        if (out.ops[i].leaf >= 0) out.leaves[out.ops[i].leaf].op = (int32_t)i;
    // inline geometry and material into every leaf record
    for (DevLeaf &lf : out.leaves) {
        if (lf.op >= 0 && lf.op < (int32_t)out.ops.size())
            for (int k = 0; k < 6; ++k) lf.g[k] = out.ops[lf.op].d[k];
        lf.texture = -1;
        lf.tex_flags = 0;
        lf.mat_kind = PT_M_DUMMY;
        if (lf.material < 0 || lf.material >= scene->n_materials) continue;
        const pt_material &m = scene->materials[lf.material];
        lf.mat_kind = m.kind;
        lf.emissive = m.emissive;
        lf.ref_idx = m.ref_idx;
        lf.fuzz = m.fuzz;
        for (int k = 0; k < 3; ++k) lf.albedo[k] = m.albedo[k];
        if (m.kind == PT_M_LAMBERTIAN || m.kind == PT_M_DIFFUSE_LIGHT || m.kind == PT_M_ISOTROPIC) {
            const pt_texture &t = scene->textures[m.texture];
            if (t.kind == PT_T_CONSTANT) for (int k = 0; k < 3; ++k) lf.albedo[k] = t.colour[k];
            else lf.texture = m.texture;
            if (t.kind == PT_T_NOISE) lf.tex_flags |= kLeafTexNoise;
            // an image texture, directly or under checkers, is what reads the surface (u, v)
            std::vector<int> todo{m.texture};
            for (size_t k = 0; k < todo.size() && k < 64; ++k) {
                const pt_texture &c = scene->textures[todo[k]];
                if (c.kind == PT_T_IMAGE) lf.tex_flags |= kLeafTexUv;
                if (c.kind == PT_T_CHECKER) {
                    if (c.even >= 0 && c.even < scene->n_textures) todo.push_back(c.even);
                    if (c.odd >= 0 && c.odd < scene->n_textures) todo.push_back(c.odd);
                }
            }
        }
    }
    return PT_OK;
}

} // namespace ptdev

extern "C" int pt_scene_flatten_info(const pt_scene_desc *scene, int32_t *n_ops, int32_t *n_leaves, int32_t *n_xforms,
                                     int32_t *max_frame_depth) {
    ptdev::FlatScene flat;
    std::string err;
    int rc = ptdev::flatten_scene(scene, flat, err);
    if (rc != PT_OK) return rc;
    if (n_ops) *n_ops = flat.n_traversal_ops;
    if (n_leaves) *n_leaves = (int32_t)flat.leaves.size();
    if (n_xforms) *n_xforms = (int32_t)flat.xforms.size();
    if (max_frame_depth) *max_frame_depth = flat.max_frame_depth;
    return PT_OK;
}
