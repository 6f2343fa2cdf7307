#!/usr/bin/env python3
"""Large-configuration sanity run on one GPU: builds configuration N (4 or 5) at a given mesh detail, reports build / upload /
render times and checks a small window of the frame against the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402
import oracle_lib  # noqa: E402
from parity import compare_frames  # noqa: E402

number = int(sys.argv[1]); detail = float(sys.argv[2]); w, h = int(sys.argv[3]), int(sys.argv[4])
t0 = time.time()
spec, cfg, _, _ = scenes.config_scene(number, detail=detail)
built = scenes.build_scene(spec)
t1 = time.time()
r = binding.Renderer(0)
r.upload(built.desc); r.set_state(built.camera, cfg)
t2 = time.time()
info = r.scene_info()
print(f"config {number}: {info.modelCount} models, {info.triangleCount} triangles, {info.nodeCount} nodes, {info.deviceBytes / 1e6:.0f} MB; "
      f"host build {t1 - t0:.1f} s, flatten+upload {t2 - t1:.1f} s", flush=True)
rgb, ids, t, st = r.render_ids(w, h)
for _ in range(3):
    _, st2 = r.render(w, h)
print(f"{w}x{h}: {st.rays} rays, {st2.gpu_ms:.3f} ms/frame, {st.rays / st2.gpu_ms / 1e3:.0f} Mrays/s; node tests/ray {st.node_tests / st.rays:.1f}, "
      f"tri tests/ray {st.tri_tests / st.rays:.1f}, gate tests/ray {st.gate_tests / st.rays:.1f}", flush=True)
win = (w // 2 - 24, h // 2 - 8, w // 2 + 24, h // 2 + 8)
o_rgb, o_ids, o_t, _, _ = oracle_lib.Oracle().render(built.desc, built.camera, cfg, w, h, window=win, threads=16)
print("window parity:", compare_frames(rgb, ids, t, o_rgb, o_ids, o_t, win), flush=True)
if len(sys.argv) > 5:
    from PIL import Image
    Image.fromarray(rgb).resize((min(w, 1920), min(h, 1080))).save(sys.argv[5])
