"""The N>1 exchange rehearsed on CPU: two gloo ranks cut their tiles out of a known frame in the layout the render
kernel writes, gather to rank 0, reassemble, and must reproduce the frame byte for byte."""
import os
import socket

import numpy as np
import pytest

from conftest import load_package

load_package()
from cuda_ray_tracer_amd import sharding  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, w, h, q, deal=(1, 1)):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)      # same on every rank (same seed)
    mine = torch.from_numpy(sharding.tile_numpy(frame, rank, world, deal))
    assert mine.numel() == sharding.tile_buffer_bytes(w, h, world, deal)
    gathered = sharding.gather_tiles(mine, rank, world, dist)
    dist.barrier()
    if rank == 0:
        out = sharding.untile_numpy(gathered.numpy(), w, h, world, deal=deal)
        q.put(bool(np.array_equal(out, frame)))
    dist.destroy_process_group()


@pytest.mark.parametrize("size,deal", [((200, 136), (1, 1)), ((64, 64), (1, 1)), ((129, 65), (1, 1)), ((200, 136), (2, 3)), ((333, 201), (1, 4))])
def test_two_rank_gather_reassembles_frame(size, deal):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, size[0], size[1], q, deal)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _batch_worker(rank, world, port, w, h, frames, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)
    batch = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(frames)]      # same on every rank
    per = sharding.tile_buffer_bytes(w, h, world)
    mine = torch.from_numpy(np.concatenate([sharding.tile_numpy(f, rank, world) for f in batch]))     # a rank's frames back to back
    gathered = sharding.gather_tiles(mine, rank, world, dist)
    dist.barrier()
    if rank == 0:
        g = gathered.numpy()
        ok = all(np.array_equal(sharding.untile_numpy(g[j * per:], w, h, world, rank_stride=frames * per), batch[j]) for j in range(frames))
        q.put(bool(ok))
    dist.destroy_process_group()


def test_two_rank_gather_of_several_frames():
    """One gather per batch of frames, as bench.py does on several GPUs: frame j of rank r lies at r * frames * per + j * per."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, 200, 136, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_layout_roundtrip_and_sizes(pkg, world):
    from cuda_ray_tracer_amd import binding
    rng = np.random.default_rng(7)
    for (w, h) in [(1920, 1080), (333, 201), (64, 64), (65, 1)]:
        assert sharding.tile_buffer_bytes(w, h, world) == binding.tile_buffer_bytes(w, h, world)
        frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        gathered = np.concatenate([sharding.tile_numpy(frame, r, world) for r in range(world)])
        assert np.array_equal(sharding.untile_numpy(gathered, w, h, world), frame)
    # every tile has exactly one owner
    tx, ty = sharding.tile_grid(1920, 1080)
    owners = sorted(t for r in range(world) for t in sharding.owned_tiles(1920, 1080, r, world))
    assert owners == list(range(tx * ty))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("deal", [(4, 5), (1, 3), (3, 3), (2, 7)])
def test_dealt_layout_roundtrip_and_sizes(pkg, world, deal):
    """Tiles dealt in periods (rank 0 takes fewer slots than the others): every tile has one owner, a rank's tiles lie in frame
    order, the library's buffer size is the numpy statement's, and cutting + reassembling reproduces the frame."""
    from cuda_ray_tracer_amd import binding
    rng = np.random.default_rng(11)
    for (w, h) in [(1920, 1080), (333, 201), (64, 64), (65, 1)]:
        assert sharding.tile_buffer_bytes(w, h, world, deal) == binding.tile_buffer_bytes(w, h, world, deal)
        frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        gathered = np.concatenate([sharding.tile_numpy(frame, r, world, deal) for r in range(world)])
        assert np.array_equal(sharding.untile_numpy(gathered, w, h, world, deal=deal), frame)
    tx, ty = sharding.tile_grid(1920, 1080)
    per_rank = [sharding.owned_tiles(1920, 1080, r, world, deal) for r in range(world)]
    assert sorted(t for tiles in per_rank for t in tiles) == list(range(tx * ty))
    assert all(tiles == sorted(tiles) for tiles in per_rank)
    if world > 1:
        k0, k1 = deal
        assert abs(len(per_rank[0]) / (tx * ty) - k0 / (k0 + (world - 1) * k1)) < 0.01


def test_balanced_deal_gives_rank_zero_less_by_the_reassembly_cost():
    assert sharding.balanced_deal(1, 0.05) == (1, 1)
    assert sharding.balanced_deal(8, 0.0) == (1, 1)
    for world, cost in [(2, 0.1), (4, 0.03), (8, 0.027), (8, 0.05)]:
        k0, k1 = sharding.balanced_deal(world, cost)
        w0 = k0 / (k0 + (world - 1) * k1)
        other = (1 - w0) / (world - 1)
        assert k0 <= k1
        assert abs((w0 + cost) - other) <= 0.06 * other       # rank 0's share + reassembly is what another rank renders
