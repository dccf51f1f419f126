// Device layout of a flattened scene. Shared by the host flattener and the HIP kernels.
//
// The reference walks a tree of Arc<dyn Hitable> recursively, always left child first
// (src/bvh.rs:74-99). Because that order is fixed, the tree is stored here as a *threaded* pre-order
// op stream: every BVH node becomes a BOX op that either falls through to the next op (its left
// subtree) or jumps to `skip` (the first op after its subtree). Traversal therefore needs no stack,
// visits primitives in exactly the reference's order, and keeps the reference's "shrinking t_max".
//
// Wrapper nodes (FlipNormals / Translate / RotateY, src/hitable.rs:80-216) are not ops: they are folded
// into per-leaf transform chains that are applied op by op, in the wrappers' nesting order, so the
// arithmetic matches the reference's nested calls. A wrapped BVH subtree becomes ENTER ... EXIT.
#pragma once

#include <stdint.h>

namespace ptdev {

enum OpKind : int32_t {
    OP_BOX = 0,     // d = min xyz, max xyz ; skip = op to jump to on a miss
    OP_SPHERE = 1,  // d = center xyz, radius
    OP_RECT = 2,    // d = amin amax bmin bmax c ; axis
    OP_BOXPRIM = 3, // d = pmin xyz, pmax xyz (the six faces of AxisAlignedBox, src/axis_aligned_box.rs:17-22)
    OP_MSPHERE = 4, // d = center0 xyz, radius, time0, time_range ; leaf.e = center_range xyz
    OP_MEDIUM = 5,  // d[0] = density ; skip = index of the first boundary op (stored after the traversal ops), d[1] = how many:
                    // one (wrapped) primitive, or a threaded sub-stream of BOX and primitive ops (a list or tree as boundary)
    OP_ENTER = 6,   // enter a transformed subtree: apply chain [chain_start, +chain_count) to the current ray
    OP_EXIT = 7,    // leave it: current ray = world ray transformed by chain [chain_start, +chain_count)
    OP_XFORM = 8    // fused list streams only: d = one DevXform step (tx ty tz s c sign) for the next primitive op
};

struct DevOp { // 64 bytes
    double d[6];
    int32_t kind;        // OpKind | axis << 8
    int32_t skip;
    int32_t leaf;        // leaf id for primitive ops, -1 otherwise
    int32_t chain_start; // first transform of the chain this op applies (local to the current frame)
    // chain_count lives in the top byte of `kind` to keep the record at 64 bytes
};
static_assert(sizeof(DevOp) == 64, "DevOp must be 64 bytes");

inline int op_kind(int32_t k) { return k & 0xff; }
inline int op_axis(int32_t k) { return (k >> 8) & 0x3; }
inline int op_chain_count(int32_t k) { return (k >> 16) & 0xff; }
inline int32_t op_pack(int kind, int axis, int chain_count) { return kind | (axis << 8) | (chain_count << 16); }

// One transform step = Translate followed by RotateY followed by a normal flip, any of which may be the
// identity (t = 0, sin = 0 / cos = 1, sign = +1). Applying an identity part is exact in f64 for finite
// values (x - 0, 1 * x - 0 * z), so a wrapper chain keeps the reference's arithmetic while the device code
// stays branch-free. The flattener merges a Translate directly followed by a RotateY into one step.
struct DevXform { // 48 bytes
    double tx, ty, tz; // Translate (src/hitable.rs:107-138)
    double s, c;       // RotateY sin / cos (src/hitable.rs:140-216)
    double sign;       // -1 when an odd number of FlipNormals (src/hitable.rs:80-105) sit at this step
};
static_assert(sizeof(DevXform) == 48, "DevXform must be 48 bytes");

// Everything shading a hit needs, in one record, so that a hit id leads to a single dependent fetch:
// the primitive's geometry (a copy of its op), its wrapper chain and its material with constant textures
// already resolved to a colour.
constexpr int32_t kLeafTexUv = 1, kLeafTexNoise = 2;
struct DevLeaf { // 192 bytes
    int32_t kind;        // OpKind of the geometry
    int32_t axis;
    int32_t mat_kind;    // pt_material_kind
    int32_t texture;     // texture index when the material's texture is not constant, else -1
    int32_t full_start;  // full chain world -> leaf space
    int32_t full_count;
    int32_t material;    // material index, reported in pt_hit
    int32_t op;          // op holding the geometry in the traversal stream
    int32_t prim[6];     // graph node reported as pt_hit.prim ([0] unless BOXPRIM, where it is per face)
    int32_t tex_flags;   // kLeafTexUv: the texture reads the surface (u, v); kLeafTexNoise: it is a NOISE texture itself
    int32_t pad;
    double g[6];         // geometry parameters (same as DevOp::d)
    double e[3];         // MSPHERE: center_range
    double albedo[3];    // constant texture colour (LAMBERTIAN / DIFFUSE_LIGHT / ISOTROPIC) or METAL albedo
    double emissive;     // LAMBERTIAN
    double ref_idx;      // DIELECTRIC
    double fuzz;         // METAL
    double pad2;
};
static_assert(sizeof(DevLeaf) == 192, "DevLeaf must be 192 bytes");

// The compact record of a scene made of plain spheres alone (no wrapper; constant colours and checkers): 64 bytes that serve BOTH
// the leaf test of the tree traversal and the shaded vertex, in the tree's leaf order, instead of a 64-byte DevOp for
// the test and a 192-byte DevLeaf somewhere else in memory for the vertex. One million spheres: 64 MB beside 62 MB of
// nodes, where ops + leaves + nodes were 318 MB (more than the 256-MB last-level cache holds). The material kind sits
// in two sign bits that carry no other information: radius and param are non-negative in every scene that qualifies
// (the host checks), and the sphere test reads the radius only squared.
//   kind - 1 = sign(radius) | sign(param) << 1   with pt_material_kind LAMBERTIAN 1, DIFFUSE_LIGHT 2, METAL 3, DIELECTRIC 4
//   param    = emissive (LAMBERTIAN), fuzz (METAL), ref_idx (DIELECTRIC), 0 (DIFFUSE_LIGHT)
struct DevSphere { // 64 bytes
    double c[3];
    double radius;
    double albedo[3]; // the constant texture's colour, or METAL's albedo; {texture index, NaN, 0} for a texture evaluated
                      // from the hit point alone (a checker)
    double param;
};
static_assert(sizeof(DevSphere) == 64, "DevSphere must be 64 bytes");

enum LightKind : int32_t { LIGHT_OTHER = 0, LIGHT_SPHERE = 1, LIGHT_RECT = 2 };
struct DevLight { // 48 bytes: a shape of the light list (src/trace.rs:123-128)
    int32_t kind;
    int32_t axis;
    double d[5]; // SPHERE: center xyz, radius ; RECT: amin amax bmin bmax c
};
static_assert(sizeof(DevLight) == 48, "DevLight must be 48 bytes");

} // namespace ptdev
