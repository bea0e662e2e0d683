"""Sort-first split of ONE frame over ranks (one process per GPU): rank r draws a band of
whole rows.  Every span of the scanline fill lives in one row and the depth test is per
pixel (reference GraphicsPoly.cpp:124-200, 340-380), so bands do not interact once the
edges are set up; clipping must still refer to the whole screen, which is why each rank
sets up the whole primitive stream and only restricts the rows it draws.  The one exchange
is the gather of the finished bands: a band is a run of whole rows, i.e. one contiguous
slice of the row-major frame, so rank 0 receives every band straight into its place with
one point-to-point transfer per peer (one xGMI link each; no ring).

Routed set-up (`share`, `lists_by_band`, `exchange_lists`): instead of every rank setting up
every polygon, rank k finds the rows of its share of the polygons (lt_polygon_rows_dev: the
projection only), the ranks exchange polygon indices by band -- again point to point, a
few bytes per polygon -- and each rank submits only the polygons that can touch its band,
in their original order (draw order decides between equal depths).
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


def share(rank: int, world: int, n: int) -> Tuple[int, int]:
    """Polygons [lo, hi) whose rows rank `rank` works out in the routing step: contiguous,
    in rank order, so that lists put together in rank order are in polygon order."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def lists_by_band(rows, first_index: int, world: int, height: int, align: int = 32):
    """rows: int tensor [n][2], {first row, last row} of polygons first_index .. first_index+n-1
    ({1, 0}: draws nothing), as lt_polygon_rows_dev gives them.  Returns one int32 tensor per
    rank r: the indices, ascending, of the polygons that can touch r's band."""
    import torch
    out = []
    live = rows[:, 0] <= rows[:, 1]
    for r in range(world):
        y0, y1 = band_rows(r, world, height, align)
        if y0 >= y1:
            out.append(torch.empty(0, dtype=torch.int32, device=rows.device))
            continue
        hit = live & (rows[:, 0] < y1) & (rows[:, 1] >= y0)
        out.append((torch.nonzero(hit).flatten() + first_index).to(torch.int32))
    return out


def exchange_lists(lists, rank: int, world: int, group=None):
    """lists[r] (int32, 1-D) goes to rank r.  Returns what the ranks sent to this one, put
    together in rank order: with rank k routing the polygons of share(k, ...) that is the
    ascending list of all polygons that can touch this rank's band.  One send and one
    receive per peer, sizes first (an all-gather of `world` counts per rank)."""
    import torch
    if world == 1:
        return lists[0]
    import torch.distributed as dist
    dev = lists[0].device
    mine = torch.tensor([int(l.numel()) for l in lists], dtype=torch.int64, device=dev)
    table = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(table, mine, group=group)
    incoming = [int(table[k][rank]) for k in range(world)]       # from rank k to this one
    parts = [torch.empty(n, dtype=torch.int32, device=dev) for n in incoming]
    parts[rank] = lists[rank]
    ops = []
    for k in range(world):
        if k == rank:
            continue
        if lists[k].numel():
            ops.append(dist.P2POp(dist.isend, lists[k].contiguous(), k, group))
        if incoming[k]:
            ops.append(dist.P2POp(dist.irecv, parts[k], k, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return torch.cat(parts)
