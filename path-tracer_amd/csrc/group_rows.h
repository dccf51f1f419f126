// Row arithmetic of the interleaved split between the members of a group (pt_group_*): member r of `world` traces the
// image rows r, r + world, r + 2 world, ...; the root's staging buffer holds the other members' rows one member behind
// the other, in rank order. Shared by the gather's host code, its two kernels and pt_group_layout, so that the host
// test of the layout exercises the arithmetic the kernels run.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HOST_DEVICE __host__ __device__
#else
#define PT_HOST_DEVICE
#endif

namespace ptdev {

// rows of an image of `height` rows that member `rank` traces
PT_HOST_DEVICE inline uint32_t group_rows_of(uint32_t height, uint32_t world, uint32_t rank) {
    return rank < height ? (height - rank + world - 1u) / world : 0u;
}
// rows that lie before member `rank`'s in the root's staging buffer (the root's own rows are never staged)
PT_HOST_DEVICE inline uint64_t group_rows_before(uint32_t height, uint32_t world, uint32_t root, uint32_t rank) {
    uint64_t before = 0;
    for (uint32_t q = 0; q < rank; ++q)
        if (q != root) before += group_rows_of(height, world, q);
    return before;
}
// image row held by row k of member `rank`'s packed rows
PT_HOST_DEVICE inline uint32_t group_image_row(uint32_t world, uint32_t rank, uint32_t k) { return rank + k * world; }
// row of the root's staging buffer that holds image row y; y must not be one of the root's own rows
PT_HOST_DEVICE inline uint64_t group_staged_row(uint32_t height, uint32_t world, uint32_t root, uint32_t y) {
    return group_rows_before(height, world, root, y % world) + y / world;
}

} // namespace ptdev
