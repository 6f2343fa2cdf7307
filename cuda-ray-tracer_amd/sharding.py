"""Image-tile sharding of one frame over the GPUs of a node.

Tile i (row-major over the grid of 32x32 tiles) belongs to rank i % world.  Each rank renders its tiles into a
compact tile-major buffer (rt_render_tiles_device), the buffers are gathered to rank 0 with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" on CPU tensors in tests), and rank 0 reassembles the frame
(rt_untile_device).  The numpy functions below state the same layout on the host; tests use them to check the
device kernels and to rehearse the exchange without a GPU.
"""
import numpy as np

TILE = 32


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_tiles(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return list(range(rank, tx * ty, world))


def tile_buffer_bytes(width, height, world):
    tx, ty = tile_grid(width, height)
    per_rank = (tx * ty + world - 1) // world
    return per_rank * TILE * TILE * 3


def tile_numpy(frame, rank, world):
    """Cuts rank's tiles out of a full [h, w, 3] uint8 frame into the compact layout the render kernel writes."""
    h, w, _ = frame.shape
    tx, _ = tile_grid(w, h)
    out = np.zeros(tile_buffer_bytes(w, h, world), np.uint8)
    view = out.reshape(-1, TILE, TILE, 3)
    for j, t in enumerate(owned_tiles(w, h, rank, world)):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE]
        view[j, :blk.shape[0], :blk.shape[1]] = blk
    return out


def untile_numpy(gathered, width, height, world, rank_stride=None):
    """Inverse of tile_numpy over `world` buffers in rank order: laid end to end, or `rank_stride` bytes apart
    (rt_untile_strided_device: one gather moved the buffers of several frames, a rank's frames lie back to back)."""
    tx, ty = tile_grid(width, height)
    per = tile_buffer_bytes(width, height, world)
    stride = per if rank_stride is None else rank_stride
    frame = np.zeros((height, width, 3), np.uint8)
    for rank in range(world):
        view = np.asarray(gathered[rank * stride:rank * stride + per]).reshape(-1, TILE, TILE, 3)
        for j, t in enumerate(owned_tiles(width, height, rank, world)):
            x0, y0 = (t % tx) * TILE, (t // tx) * TILE
            hh, ww = min(TILE, height - y0), min(TILE, width - x0)
            frame[y0:y0 + hh, x0:x0 + ww] = view[j, :hh, :ww]
    return frame


def gather_tiles(tiles, rank, world, dist, gathered=None):
    """Gathers every rank's tile buffer (a 1-D uint8 torch tensor) to rank 0.  Returns the concatenated buffer on
    rank 0 (allocated once by the caller and passed as `gathered` to avoid per-frame allocation), None elsewhere."""
    import torch
    if world == 1:
        return tiles
    if rank == 0:
        if gathered is None:
            gathered = torch.empty(tiles.numel() * world, dtype=torch.uint8, device=tiles.device)
        dist.gather(tiles, list(gathered.chunk(world)), dst=0)
        return gathered
    dist.gather(tiles, None, dst=0)
    return None
