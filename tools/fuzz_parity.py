#!/usr/bin/env python3
"""Differential campaign on the GPU box: seeded random scenes (the generator of tests/test_gpu_random_scenes.py) rendered by
the library and by the CPU oracle; prints every seed that misses the parity bar and a summary.
Usage: python tools/fuzz_parity.py <first seed> <count> [width height [variety]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402
import oracle_lib  # noqa: E402
from parity import compare_frames  # noqa: E402
from test_gpu_random_scenes import random_scene  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (224, 126)
VARIETY = len(sys.argv) > 5 and sys.argv[5] == "variety"      # tilted planes, materials without a specular part, more debug objects
r = binding.Renderer(0)
O = oracle_lib.Oracle()
bad = 0; lsb = 0; total_px = 0; mesh_px = 0
for seed in range(first, first + count):
    spec, cfg = random_scene(seed, variety=VARIETY)
    built = scenes.build_scene(spec, build_on_device=bool(seed % 3 == 2))
    r.upload(built.desc); r.set_state(built.camera, cfg)
    g_rgb, g_ids, g_t, st = r.render_ids(w, h)
    plain, st_plain = r.render(w, h)          # the kernel variant without id / distance outputs must paint the same frame
    if not np.array_equal(plain, g_rgb) or st_plain.rays != st.rays:
        bad += 1
        print("VARIANT MISMATCH seed", seed, int((plain != g_rgb).any(axis=-1).sum()), "pixels", flush=True)
    o_rgb, o_ids, o_t, o_rays, _ = O.render(built.desc, built.camera, cfg, w, h, threads=16)
    s = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, None)
    total_px += s["pixels"]; lsb += s["px_gt0"]; mesh_px += int((g_ids[..., 0] >> 24 == 4).sum())
    if s["px_gt1"] or s["id_mismatch"] or s["t_bit_mismatch"] or st.rays != o_rays:
        bad += 1
        print("MISMATCH seed", seed, s, "rays", st.rays, o_rays, flush=True)
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} scenes, {bad} bad", flush=True)
print(f"{count} scenes from seed {first} at {w}x{h}: {bad} miss the bar; {lsb} of {total_px} pixels differ by 1 LSB; {mesh_px} mesh pixels", flush=True)
r.close()
