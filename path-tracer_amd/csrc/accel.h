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

struct AccelNode { // 128 bytes = one L2 line: the boxes of BOTH children, so that one fetch decides both
    double lo[2][3];
    double hi[2][3];
    int32_t child[2]; // >= 0: inner node index; < 0: leaf, first primitive = ~child
    int32_t count[2]; // primitives of a leaf child (0 for an inner child)
    int32_t pad[4];
};
static_assert(sizeof(AccelNode) == 128, "AccelNode must be 128 bytes");

constexpr int kAccelMaxDepth = 32; // traversal stack entries per ray (32 KiB of LDS per 256-thread workgroup)

struct Accel {
    std::vector<AccelNode> nodes; // nodes[0] is the root
    std::vector<DevOp> prims;     // primitive ops in leaf order, each carrying its FULL world -> primitive chain
    int depth = 0;
};

// World-space bounds of primitive op `op` (an op of flat.ops with leaf >= 0). Returns false for non-primitives.
bool accel_prim_bounds(const FlatScene &flat, const DevOp &op, double lo[3], double hi[3]);

// Builds the tree over every primitive op of the traversal stream. Returns false (with a message) if the scene has
// no primitives or the tree would be deeper than kAccelMaxDepth.
bool build_accel(const FlatScene &flat, Accel &out, std::string &error);

} // namespace ptdev
