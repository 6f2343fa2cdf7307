#!/usr/bin/env python3
"""Times rt_upload_scene with the octrees built on the host and on the GPU for one configuration, and checks that both
trees render the same frame.  Usage: build_time.py <config> <detail> [width height]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

number = int(sys.argv[1]); detail = float(sys.argv[2])
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
t0 = time.perf_counter()
spec, cfg, _, _ = scenes.config_scene(number, detail=detail)
built = scenes.build_scene(spec)
print(f"scene generated and baked in {time.perf_counter() - t0:.1f} s", flush=True)
r = binding.Renderer(0)
frames = {}
for on_device in (False, True, False, True):
    built.desc.buildOnDevice = 1 if on_device else 0
    t1 = time.perf_counter()
    r.upload(built.desc)
    dt = time.perf_counter() - t1
    r.set_state(built.camera, cfg)
    info = r.scene_info()
    rgb, st = r.render(w, h)
    for _ in range(3):
        _, st = r.render(w, h)
    frames[on_device] = rgb
    print(f"{'GPU ' if on_device else 'host'} build: upload {dt * 1e3:8.1f} ms  ({info.triangleCount} triangles, {info.nodeCount} nodes, "
          f"{info.leafCount} leaves, depth {info.maxDepth}); render {st.gpu_ms:.3f} ms/frame at {w}x{h}", flush=True)
print("frames identical:", bool(np.array_equal(frames[False], frames[True])))
