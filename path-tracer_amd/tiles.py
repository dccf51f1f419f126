"""Image partition across GPUs and the gather of the rendered rows to the presenting rank.

The reference splits the image into tile jobs for a CPU thread pool (src/lib.rs:345-394). Across GPUs two splits
are offered, both into disjoint sets of scanlines, so that assembling the image is a gather, not a reduction:

* interleaved rows (what bench.py uses): rank r traces rows r, r + N, r + 2N, ... in ONE pt_render call
  (pt_render_args.row_step / row_phase). Neighbouring rows cost the same, so the load is even to within a percent,
  and every rank keeps one large pass. Measured on the Cornell box: 64-row stripes are 14 % out of balance at 8 GPUs.
* stripes of `rows` scanlines dealt round-robin (the reference's rectangles, one pt_render call per stripe).
"""
from __future__ import annotations

STRIPE_ROWS = 64


def stripes_for(rank: int, world: int, height: int, rows: int = STRIPE_ROWS):
    """(y0, y1) row ranges (y counted from the bottom, as in the image buffer) rendered by `rank`."""
    n_stripes = (height + rows - 1) // rows
    return [(s * rows, min((s + 1) * rows, height)) for s in range(n_stripes) if s % world == rank]


def max_stripes(world: int, height: int, rows: int = STRIPE_ROWS) -> int:
    return len(stripes_for(0, world, height, rows))


def gather_stripes(image, rank: int, world: int, dist, rows: int = STRIPE_ROWS, pack=None, gather_list=None):
    """Collects every rank's stripes of `image` ([H, W, 4] tensor, any device) into rank 0's `image`.

    One collective: each rank packs its stripes into a [max_stripes, rows, W, 4] tensor, rank 0 receives them with
    dist.gather (RCCL when the process group is "nccl") and scatters them into place.
    """
    import torch

    height, width = image.shape[0], image.shape[1]
    mine = stripes_for(rank, world, height, rows)
    if pack is None:
        pack = torch.zeros((max_stripes(world, height, rows), rows, width, 4), dtype=image.dtype, device=image.device)
    if rank == 0 and gather_list is None:
        gather_list = [torch.zeros_like(pack) for _ in range(world)]
    for k, (y0, y1) in enumerate(mine):
        pack[k, : y1 - y0].copy_(image[y0:y1])
    dist.gather(pack, gather_list if rank == 0 else None, dst=0)
    if rank == 0:
        for r in range(1, world):
            for k, (y0, y1) in enumerate(stripes_for(r, world, height, rows)):
                image[y0:y1].copy_(gather_list[r][k, : y1 - y0])
    return image


def rows_for(rank: int, world: int):
    """(row_step, row_phase) of pt_render_args for the interleaved split."""
    return world, rank


def max_rows(world: int, height: int) -> int:
    return (height + world - 1) // world


def gather_rows(image, rank: int, world: int, dist, pack=None, gather_list=None):
    """Collects the interleaved rows of every rank into rank 0's `image` ([H, W, 4] tensor) with one dist.gather."""
    import torch

    height, width = image.shape[0], image.shape[1]
    if pack is None:
        pack = torch.zeros((max_rows(world, height), width, 4), dtype=image.dtype, device=image.device)
    if rank == 0 and gather_list is None:
        gather_list = [torch.zeros_like(pack) for _ in range(world)]
    mine = image[rank::world]
    pack[: mine.shape[0]].copy_(mine)
    dist.gather(pack, gather_list if rank == 0 else None, dst=0)
    if rank == 0:
        for r in range(1, world):
            dst = image[r::world]
            dst.copy_(gather_list[r][: dst.shape[0]])
    return image
