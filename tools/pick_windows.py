#!/usr/bin/env python3
"""Prints candidate oracle windows (48 x 24 pixels, partly covered by meshes) for poses of the camera path of a benchmark
configuration at its full frame size: python tools/pick_windows.py <config> <detail> <pose> [<pose> ...]   (needs the GPU)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402


def main():
    config, detail = int(sys.argv[1]), float(sys.argv[2])
    poses = [int(a) for a in sys.argv[3:]]
    t0 = time.time()
    spec, cfg, w, h = scenes.config_scene(config, detail=detail)
    built = scenes.build_scene(spec)
    print("scene built in", round(time.time() - t0, 1), "s", flush=True)
    r = binding.Renderer(0)
    t0 = time.time()
    r.upload(built.desc)
    print("uploaded in", round(time.time() - t0, 1), "s,", r.scene_info().triangleCount, "triangles", flush=True)
    for k in poses:
        pos, yaw, pitch = scenes.camera_path_pose(spec, k)
        r.set_state(binding.camera_look(pos, spec.fov, yaw, pitch), cfg)
        _, ids, t, st = r.render_ids(w, h)
        mesh = (ids[..., 0] >> 24) == 4
        cover = mesh.reshape(h // 24, 24, w // 48, 48).mean(axis=(1, 3))
        ys, xs = np.nonzero((cover > 0.3) & (cover < 0.9))
        tm = np.where(mesh, t, np.nan)
        picks = []
        for q in (0.05, 0.5, 0.95):          # nearest, middle, farthest partly covered windows by hit distance
            d = np.array([np.nanmean(tm[y * 24:(y + 1) * 24, x * 48:(x + 1) * 48]) for y, x in zip(ys, xs)])
            i = int(np.argsort(d)[int(q * (len(d) - 1))])
            picks.append((int(xs[i]) * 48, int(ys[i]) * 24, int(xs[i]) * 48 + 48, int(ys[i]) * 24 + 24))
        print("pose", k, "mesh fraction", float(mesh.mean()), "rays", st.rays, "windows", picks, flush=True)
    r.close()


if __name__ == "__main__":
    main()
