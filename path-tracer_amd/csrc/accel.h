// Fast-mode acceleration structure: a binary SAH BVH over the world-space boxes of the flattened scene's primitive
// ops, built on the host at upload. It replaces, for speed only, the traversal of the reference's own tree
// (src/bvh.rs:13-99: random split axis, children always visited left first): the closest hit over the same set of
// primitives is the same whichever tree finds it, up to exact ties. Exact mode never uses it.
#pragma once

#include "device_scene.h"
#include "flatten.h"

#include <string>
#include <vector>

namespace ptdev {

// 64 bytes: the boxes of BOTH children, so that one fetch decides both. The tree of a million primitives does not
// fit the 4 MiB L2 of an XCD, so traversal speed is bytes per node visit. Boxes are stored in f32, rounded OUTWARD
// (never smaller than the f64 box), and the slab test runs in f64 on the widened values with the f64 ray: no
// rounding analysis is needed for it to be conservative, and the primitive tests stay f64.
struct AccelNode {
    float lo[2][3];
    float hi[2][3];
    int32_t child[2]; // >= 0: inner node index; < 0: leaf, ~child = (first primitive << 3) | count, count <= 7
    int32_t pad[2];
};
static_assert(sizeof(AccelNode) == 64, "AccelNode must be 64 bytes");
inline int32_t accel_leaf_ref(int32_t first, int32_t count) { return ~((first << 3) | count); }

// The same node in 32 bytes, for the extend stage over big trees and for the pass kernel of scenes of plain spheres (trees
// of any size): the planes of both child boxes as 16-bit steps of a grid over the scene's box, or over the tree beside
// the root's leaf when a small tree's root holds one (lower planes rounded down and upper planes up, by one step more
// than needed, so that the decoded box always contains the f32 one). A node visit is then two 16-byte loads per lane instead of four. That
// matters because the vector L1 serves every lane's 16 bytes of a divergent load as an access of its own and is the
// unit that saturates on the million-sphere tree; slab tests only cull, so a slightly larger box changes no hit.
struct AccelNodeQ {
    uint16_t lo[3][2]; // [axis][child]
    uint16_t hi[3][2];
    int32_t child[2];
};
static_assert(sizeof(AccelNodeQ) == 32, "AccelNodeQ must be 32 bytes");
struct AccelGrid {
    double gmin[3], cell[3]; // plane = gmin + steps * cell
};
constexpr int32_t kAccelQuantFrom = 32768; // trees of at least this many nodes are also kept in the 32-byte form (trees over compact sphere records: always)

constexpr int kAccelMaxDepth = 28; // traversal stack entries per ray (32 KiB of LDS per 256-thread workgroup)

struct Accel {
    std::vector<AccelNode> nodes; // nodes[0] is the root
    std::vector<DevOp> prims;     // primitive ops in leaf order, each carrying its FULL world -> primitive chain
    // the same primitives as compact records (DevSphere), when every one of them qualifies; else empty
    std::vector<DevSphere> spheres;
    int depth = 0;
    // for refitting on the device: where a node's own bounds are stored (parent * 2 + slot, -1 for the root) and
    // the nodes grouped by depth, deepest level first (level_start has one more entry than there are levels)
    std::vector<int32_t> parent_slot;
    std::vector<int32_t> by_level;
    std::vector<int32_t> level_start;
};

// World-space bounds of primitive op `op` (an op of flat.ops with leaf >= 0). Returns false for non-primitives.
bool accel_prim_bounds(const FlatScene &flat, const DevOp &op, double lo[3], double hi[3]);

// Builds the tree over every primitive op of the traversal stream. Returns false (with a message) if the scene has
// no primitives or the tree would be deeper than kAccelMaxDepth.
bool build_accel(const FlatScene &flat, Accel &out, std::string &error);

// Fills parent_slot / by_level / level_start from nodes.
void accel_link_levels(Accel &out);

} // namespace ptdev
