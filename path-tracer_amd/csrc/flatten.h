// Host-side flattener: scene graph (include/ptgpu.h wire format) -> threaded op stream (device_scene.h).
#pragma once

#include "../../include/ptgpu.h"
#include "device_scene.h"

#include <string>
#include <vector>

namespace ptdev {

struct FlatScene {
    std::vector<DevOp> ops;     // [0, n_traversal_ops) is the traversal stream; boundary ops of media follow
    int32_t n_traversal_ops = 0;
    std::vector<DevXform> xforms;
    std::vector<DevLeaf> leaves;
    std::vector<DevLight> lights;
    int32_t max_frame_depth = 0; // nesting depth of transformed subtrees
    bool has_medium = false;
    bool has_frames = false;
    // some medium's boundary is more than one (wrapped) primitive: a list or a tree of them, stored as a threaded
    // sub-stream behind the traversal stream (DevOp::d[1] of the MEDIUM op holds its length)
    bool general_boundaries = false;
};

// Returns PT_OK or PT_ERR_UNSUPPORTED_SCENE / PT_ERR_INVALID_ARG with a message in `error`.
int flatten_scene(const pt_scene_desc *scene, FlatScene &out, std::string &error);

} // namespace ptdev
