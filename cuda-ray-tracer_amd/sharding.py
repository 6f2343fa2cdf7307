"""Image-tile sharding of one frame over the GPUs of a node.

Tile i (row-major over the grid of 32x32 tiles) belongs to rank i % world - or, with a deal (k0, k1) other than (1, 1),
tiles are dealt in periods of k0 + (world - 1) * k1 consecutive tiles of which the first k0 go to rank 0 and k1 each to
the others (rt_set_tile_deal: rank 0 also reassembles every frame, so it renders less).  Each rank renders its tiles into a
compact tile-major buffer (rt_render_tiles_device), the buffers are gathered to rank 0 with torch.distributed
(backend "nccl" = RCCL over xGMI on the GPUs; "gloo" on CPU tensors in tests), and rank 0 reassembles the frame
(rt_untile_device).  The numpy functions below state the same layout on the host; tests use them to check the
device kernels and to rehearse the exchange without a GPU.
"""
import numpy as np

TILE = 32


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_tiles(width, height, rank, world, deal=(1, 1)):
    """The tiles of `rank`, in the order in which they lie in its buffer."""
    tx, ty = tile_grid(width, height)
    k0, k1 = deal
    period = k0 + (world - 1) * k1
    first, run = (0, k0) if rank == 0 else (k0 + (rank - 1) * k1, k1)
    return [t for t in range(tx * ty) if first <= t % period < first + run]


def tile_buffer_bytes(width, height, world, deal=(1, 1)):
    """Every rank's buffer has the size of the largest share."""
    return max(len(owned_tiles(width, height, r, world, deal)) for r in range(min(world, 2))) * TILE * TILE * 3


def balanced_deal(world, reassembly_share):
    """A deal (k0, k1) under which rank 0's render share plus the reassembly costs about what the others' shares cost.
    reassembly_share = time rank 0 spends gathering and reassembling a frame / time one GPU needs to render the whole frame.
    With shares w0 and (1 - w0) / (world - 1): w0 + reassembly_share = (1 - w0) / (world - 1)."""
    if world <= 1:
        return (1, 1)
    w0 = max((1.0 - (world - 1) * reassembly_share) / world, 0.0)
    share = lambda k0, k1: k0 / (k0 + (world - 1) * k1)
    best = (1, 1)
    for k1 in range(1, 9):      # short periods first: the first deal within 4 % of the wanted share is taken
        for k0 in range(1, k1 + 1):
            if abs(share(k0, k1) - w0) < abs(share(*best) - w0) - 1e-12:
                best = (k0, k1)
        if abs(share(*best) - w0) <= 0.04 * w0:
            break
    return best


def tile_numpy(frame, rank, world, deal=(1, 1)):
    """Cuts rank's tiles out of a full [h, w, 3] uint8 frame into the compact layout the render kernel writes."""
    h, w, _ = frame.shape
    tx, _ = tile_grid(w, h)
    out = np.zeros(tile_buffer_bytes(w, h, world, deal), np.uint8)
    view = out.reshape(-1, TILE, TILE, 3)
    for j, t in enumerate(owned_tiles(w, h, rank, world, deal)):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE]
        view[j, :blk.shape[0], :blk.shape[1]] = blk
    return out


def untile_numpy(gathered, width, height, world, rank_stride=None, deal=(1, 1)):
    """Inverse of tile_numpy over `world` buffers in rank order: laid end to end, or `rank_stride` bytes apart
    (rt_untile_strided_device: one gather moved the buffers of several frames, a rank's frames lie back to back)."""
    tx, ty = tile_grid(width, height)
    per = tile_buffer_bytes(width, height, world, deal)
    stride = per if rank_stride is None else rank_stride
    frame = np.zeros((height, width, 3), np.uint8)
    for rank in range(world):
        view = np.asarray(gathered[rank * stride:rank * stride + per]).reshape(-1, TILE, TILE, 3)
        for j, t in enumerate(owned_tiles(width, height, rank, world, deal)):
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
