#!/usr/bin/env python3
"""Regenerates the golden frames under tests/golden/ with the CPU oracle.

The reference ships no golden images, and the file that holds its frame loop and shading (kernels.cu) does not compile
outside MSVC + nvcc, so these whole-frame vectors are outputs of the oracle (oracle/oracle.c), not of the reference itself:
they pin the oracle against regressions and let the GPU path be checked without running the oracle.  The vectors that DO
come from the reference's own compiled routines are made by make_ref_golden.py (refgen_*.npz).  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402
import oracle_lib  # noqa: E402


def cases():
    """name -> (SceneSpec, Config, width, height)"""
    out = {}
    for key, label in [("Z", "ref_Z"), ("X", "ref_X"), ("C", "ref_C"), ("V", "ref_V"), ("B", "ref_B"), ("N", "ref_N")]:
        index, cfg = scenes.preset_config(key)
        spec = scenes.reference_scene(index)
        scenes.apply_frame_fixups(spec, cfg)
        out[label] = (spec, cfg, 160, 90)
    for key, label in [("M", "ref_M"), (".", "ref_period"), (",", "ref_comma")]:
        index, cfg = scenes.preset_config(key)
        spec = scenes.reference_scene(index, detail=0.4)
        scenes.apply_frame_fixups(spec, cfg)
        out[label] = (spec, cfg, 128, 72)
    spec, cfg, _, _ = scenes.config_scene(2, detail=0.3)
    out["cfg2"] = (spec, cfg, 128, 72)
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.15)
    out["cfg3"] = (spec, cfg, 160, 90)
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.15)
    cfg.antiAliasing = 1
    out["cfg3_aa"] = (spec, cfg, 96, 54)
    index, cfg = scenes.preset_config("C")
    cfg.renderSoftShadows = 1; cfg.renderHardShadows = 0; cfg.areaLightSpecularEffect = 1
    spec = scenes.reference_scene(index)
    scenes.apply_frame_fixups(spec, cfg)
    out["ref_C_soft"] = (spec, cfg, 128, 72)
    return out


def main():
    O = oracle_lib.Oracle()
    for name, (spec, cfg, w, h) in cases().items():
        built = scenes.build_scene(spec)
        rgb, ids, t, rays, _ = O.render(built.desc, built.camera, cfg, w, h, threads=8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgb=rgb, ids=ids, t=t, rays=np.int64(rays))
        print(name, w, h, rays)


if __name__ == "__main__":
    main()
