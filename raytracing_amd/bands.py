"""Sort-first split of ONE frame over ranks (one process per GPU): rank r draws a band of
whole rows.  Every span of the scanline fill lives in one row and the depth test is per
pixel (reference GraphicsPoly.cpp:124-200, 340-380), so bands do not interact once the
edges are set up; clipping must still refer to the whole screen, which is why each rank
sets up the whole primitive stream and only restricts the rows it draws.  The one exchange
is the gather of the finished bands: a band is a run of whole rows, i.e. one contiguous
slice of the row-major frame, so rank 0 receives every band straight into its place with
one point-to-point transfer per peer (one xGMI link each; no ring).
"""
from typing import List, Tuple


def band_rows(rank: int, world: int, height: int, align: int = 32) -> Tuple[int, int]:
    """Rows [y0, y1) of rank `rank`: bands are multiples of `align` rows (the fill's tile
    height, so that no tile is shared by two ranks) and differ by at most one such unit;
    the last band takes the remainder.  A rank beyond the number of units gets (height, height)."""
    units = (height + align - 1) // align
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, height), min(u1 * align, height)


def all_bands(world: int, height: int, align: int = 32) -> List[Tuple[int, int]]:
    return [band_rows(r, world, height, align) for r in range(world)]


def gather_bands(frames, rank: int, world: int, height: int, align: int = 32, dst: int = 0, group=None):
    """frames: tensors [H][W] (colour, depth, ...), each rank's own band of rows already
    drawn in place.  After the call rank `dst` holds every band of every tensor.
    torch.distributed must be initialised when world > 1 (backend nccl = RCCL for CUDA
    tensors, gloo for CPU tensors)."""
    if world == 1:
        return
    import torch.distributed as dist
    ops = []
    for t in frames:
        assert t.is_contiguous() and t.shape[0] == height
        if rank == dst:
            for r in range(world):
                y0, y1 = band_rows(r, world, height, align)
                if r != dst and y0 < y1:
                    ops.append(dist.P2POp(dist.irecv, t[y0:y1], r, group))
        else:
            y0, y1 = band_rows(rank, world, height, align)
            if y0 < y1:
                ops.append(dist.P2POp(dist.isend, t[y0:y1], dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
