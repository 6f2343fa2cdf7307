#!/usr/bin/env python3
"""Emulates the ranks of an N-GPU run on one GPU: renders each rank's share of the benchmark frame in turn (render only,
no exchange) and prints the slowest rank's time per frame.  RT_FLIGHT=F alternates consecutive frames over F contexts
and streams; RT_CFG selects the configuration.  Usage: python tools/rank_time.py [world ...]"""
import os
import sys
import time

ROOT = os.environ.get("RT_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

cfgno = int(os.environ.get("RT_CFG", "3"))
flight = int(os.environ.get("RT_FLIGHT", "1"))
spec, cfg, w, h = scenes.config_scene(cfgno)
built = scenes.build_scene(spec)
lanes = []
for i in range(flight):
    r = binding.Renderer(0); r.upload(built.desc); r.set_state(built.camera, cfg)
    lanes.append((r, torch.cuda.current_stream() if i == 0 else torch.cuda.Stream()))
for world in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    nbytes = binding.tile_buffer_bytes(w, h, world)
    tiles = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in lanes]
    worst = 0
    for rank in range(world):
        def frames(n):
            for k in range(n):
                r, s = lanes[k % flight]
                r.render_tiles_device(w, h, rank, world, tiles[k % flight].data_ptr(), nbytes, s.cuda_stream)
        frames(20 * flight)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frames(200)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        worst = max(worst, dt)
    print(f"cfg {cfgno} tile {binding.RT_TILE} frames in flight {flight} world {world}: slowest rank {worst * 1e3:.3f} ms per frame (render only, no exchange)", flush=True)
