// Host-side launcher of the fast-mode path pool over primitive lists without media (k_path_pool<TEX, 1, false>), which lives in a
// translation unit of its own (pool_list.hip) so that it can be compiled with a scheduling strategy of its own.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_device.h"

namespace ptdev {

// tex: the scene's texture level (0, 1, 2); realtime: the pass is a realtime frame (the RT variant). `consts_value` points
// at the pass's PoolConsts on the host (the kernel's first argument: a realtime frame reads its constants from the launch
// packet), `consts` is their copy in device memory (what an offline pass reads). The other arguments are the kernel's.
void launch_path_pool_list(int tex, bool realtime, uint32_t grid, hipStream_t stream, const SceneView &sv,
                           const void *consts_value, const void *consts,
                           uint32_t n, uint32_t max_depth, double *lx, double *ly, double *lz, uint32_t *counts,
                           uint32_t flags, uint32_t refill, uint32_t chunk);

} // namespace ptdev
