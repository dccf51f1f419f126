// The fast-mode path pool over primitive lists without media, k_path_pool<TEX, 1, false> (kernels_pool.h): the kernel that
// traces the Cornell box, in a translation unit of its own so that its scheduling strategy can be chosen apart from the
// other kernels' (the Makefile's POOL_LIST_FLAGS). While the kernel read its pass constants by vector loads it sat a few
// spills above its 128 registers, and -mllvm -amdgpu-sched-strategy=iterative-minreg was worth +2 to +3 % (the same flag
// costs the tree pool 3 to 34 %); with those constants in scalar registers it has no scratch under any strategy, and the
// default scheduler is the fastest again (+1 % over iterative-minreg, max-ilp between the two). Same sources, same
// arithmetic (-ffp-contract=off in both).
#include <hip/hip_runtime.h>

#include "accel.h"
#include "pool_list.h"
#include "pt_device.h"

using namespace ptdev;

namespace {

#include "kernels_queue.h"
#include "kernels_extend.h"
#include "kernels_shade.h"
#include "kernels_pool.h"

template <int TEX>
void launch_tex(bool realtime, uint32_t grid, hipStream_t stream, const SceneView &sv, const PoolConsts &value,
                const PoolConsts *consts, uint32_t n,
                uint32_t max_depth, double *lx, double *ly, double *lz, uint32_t *counts, uint32_t flags, uint32_t refill,
                uint32_t chunk) {
    if (realtime)
        hipLaunchKernelGGL((k_path_pool<TEX, 1, false, true>), dim3(grid), dim3(kBlock), 0, stream, value, sv, consts, n, max_depth, lx,
                           ly, lz, counts, flags, refill, chunk);
    else
        hipLaunchKernelGGL((k_path_pool<TEX, 1, false, false>), dim3(grid), dim3(kBlock), 0, stream, value, sv, consts, n, max_depth, lx,
                           ly, lz, counts, flags, refill, chunk);
}

} // namespace

namespace ptdev {

void launch_path_pool_list(int tex, bool realtime, uint32_t grid, hipStream_t stream, const SceneView &sv,
                           const void *consts_value, const void *consts,
                           uint32_t n, uint32_t max_depth, double *lx, double *ly, double *lz, uint32_t *counts,
                           uint32_t flags, uint32_t refill, uint32_t chunk) {
    const PoolConsts *pc = (const PoolConsts *)consts;
    const PoolConsts &value = *(const PoolConsts *)consts_value;
    if (tex == 0) launch_tex<0>(realtime, grid, stream, sv, value, pc, n, max_depth, lx, ly, lz, counts, flags, refill, chunk);
    else if (tex == 1) launch_tex<1>(realtime, grid, stream, sv, value, pc, n, max_depth, lx, ly, lz, counts, flags, refill, chunk);
    else launch_tex<2>(realtime, grid, stream, sv, value, pc, n, max_depth, lx, ly, lz, counts, flags, refill, chunk);
}

} // namespace ptdev
