#!/usr/bin/env python3
"""Frame time of a benchmark configuration with ray types switched off one by one (hard shadows, reflections, deeper
reflection levels): what each kind of ray adds to the frame.  Usage (GPU): python tools/ray_type_cost.py [config] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

number = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
spec, cfg, w, h = scenes.config_scene(number)
built = scenes.build_scene(spec)
r = binding.Renderer(0)
r.upload(built.desc)
buf = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
variants = [("full", {}), ("no hard shadows", dict(renderHardShadows=0)), ("no reflections", dict(reflections=0)),
            ("reflection depth 1", dict(maxDepth=1)), ("primary rays only", dict(renderHardShadows=0, reflections=0))]
for name, changes in variants:
    c = scenes.config_scene(number)[1]
    for k, v in changes.items():
        setattr(c, k, v)
    r.set_state(built.camera, c)
    rays = r.count_rays(w, h).rays
    for _ in range(40):
        r.render_device(w, h, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.render_device(w, h, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / frames
    print(f"config {number} {name:22s} {ms:.4f} ms/frame  {rays} rays", flush=True)
r.close()
