"""Screen-tile-row split across the GPUs of one node.

Each rank owns a contiguous band of 32-pixel tile rows (the last rank's band may be short), renders
only that band (mirhi_device_set_tile_split) and the bands are exchanged with ONE all-gather of
the final RGBA rows -- only when the split is enabled; the 1-GPU path never touches
torch.distributed.  The frame buffer is allocated with `padded_rows` rows so every rank's slice
has the same size and the all-gather runs in place (input = output slice of this rank).
"""
from __future__ import annotations

TILE = 32


def tiles_y(height: int) -> int:
    return (height + TILE - 1) // TILE


def band_tile_rows(height: int, rank: int, world: int):
    """Same formula as band_tile_rows() in csrc/mirhi_api.hip."""
    ty = tiles_y(height)
    per = (ty + world - 1) // world
    b = min(rank * per, ty)
    e = min(b + per, ty)
    return b, e


def band_rows(height: int, rank: int, world: int):
    b, e = band_tile_rows(height, rank, world)
    return min(b * TILE, height), min(e * TILE, height)


def split_tile_rows(height: int, rank: int, world: int, layout: str = "interleaved"):
    """The tile rows `rank` of `world` rasterizes, as split_rows() in csrc/mirhi_api.hip gives them: (first, step, count)."""
    ty = tiles_y(height)
    if world <= 1:
        return 0, 1, ty
    if layout == "bands":
        b, e = band_tile_rows(height, rank, world)
        return b, 1, e - b
    return rank, world, ((ty - rank + world - 1) // world if rank < ty else 0)


def owned_pixel_rows(height: int, rank: int, world: int, layout: str = "interleaved"):
    """[(begin, end)] pixel-row runs of the frame that `rank` renders (one run per band / per owned tile row), as mirhi_comm_all_gather_bands sends them"""
    first, step, count = split_tile_rows(height, rank, world, layout)
    if step == 1:
        runs = [(first * TILE, (first + count) * TILE)]
    else:
        runs = [((first + k * step) * TILE, (first + k * step + 1) * TILE) for k in range(count)]
    return [(min(b, height), min(e, height)) for b, e in runs if min(b, height) < min(e, height)]


def rows_per_rank(height: int, world: int) -> int:
    return ((tiles_y(height) + world - 1) // world) * TILE


def padded_rows(height: int, world: int) -> int:
    return rows_per_rank(height, world) * world


def all_gather_bands(frame, rank: int, world: int, group=None, via_host: bool = False):
    """In-place all-gather of the row bands. `frame` is a (padded_rows, W, C) tensor (any device);
    rank r has rendered rows [r*per, (r+1)*per) and receives everyone else's.  `via_host` stages through CPU
    tensors (gloo rehearsals on a box without RCCL peers); the product path is the single RCCL collective."""
    import torch.distributed as dist
    per = frame.shape[0] // world
    assert per * world == frame.shape[0], "frame must have padded_rows(height, world) rows"
    mine = frame[rank * per:(rank + 1) * per]
    if frame.is_cuda and via_host:
        host = frame.cpu()
        all_gather_bands(host, rank, world, group=group)
        frame.copy_(host)
    elif frame.is_cuda:
        dist.all_gather_into_tensor(frame, mine, group=group)
    else:  # gloo (CPU tests): list form
        chunks = [frame[r * per:(r + 1) * per] for r in range(world)]
        outs = [c.clone() for c in chunks]
        dist.all_gather(outs, mine.contiguous(), group=group)
        for c, o in zip(chunks, outs):
            c.copy_(o)
    return frame


def exchange_bands_direct(frame, height: int, rank: int, world: int, group=None, layout: str = "bands"):
    """The MIRHI_GATHER_DIRECT pattern of mirhi_comm_all_gather_bands (csrc/mirhi_api.hip) on torch.distributed: every rank
    sends the rows it rendered straight to every peer and receives theirs, one batch of point-to-point transfers, in place on an UNPADDED
    (height, W, C) frame -- shares may differ in size (the last band / tile row is short when the rows do not divide); with interleaved rows
    a rank's share is one piece per tile row it owns."""
    import torch.distributed as dist
    mine = owned_pixel_rows(height, rank, world, layout)
    ops = []
    for r in range(world):
        if r == rank:
            continue
        for b0, e0 in mine:
            ops.append(dist.P2POp(dist.isend, frame[b0:e0], r, group=group))
        for b, e in owned_pixel_rows(height, r, world, layout):
            ops.append(dist.P2POp(dist.irecv, frame[b:e], r, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return frame
