#!/usr/bin/env python3
"""Times the frame reassembly (rt_untile_batch_device) alone on the GPU: microseconds per frame and bytes read + written per
second for 1080p, 4K and 8K frames, 8 ranks x 16 frames and 2 ranks x 4 frames per batch.  Usage: python tools/untile_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_package
load_package()
import torch
from cuda_ray_tracer_amd import binding
r = binding.Renderer(0)
for (w, h) in ((1920, 1080), (3840, 2160), (7680, 4320)):
    for world, batch in ((8, 16), (2, 4)):
        nb = binding.tile_buffer_bytes(w, h, world)
        g = torch.randint(0, 255, (batch * world * nb,), dtype=torch.uint8, device="cuda")
        out = torch.empty(batch * w * h * 3, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        def go():
            r.untile_batch_device(w, h, world, g.data_ptr(), batch * nb, nb, batch, out.data_ptr(), s)
        for _ in range(3): go()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): go()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{w}x{h} world {world} batch {batch}: {dt*1e6/batch:.2f} us per frame, {2*w*h*3*batch/dt/1e12:.2f} TB/s", flush=True)
