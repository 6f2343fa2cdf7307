#!/usr/bin/env python3
"""Diagnostic for a build whose instrumented kernel counts, in profile slot 6, the traversal turns that start with entries in
both pools (and those where both pools together fit one turn): how many turns a combined node + triangle turn could save.
The counting lines are added with tools/build_variant.sh (VARIANT_SED); run with RT_LIB_OVERRIDE pointing at that build."""
import os, sys
import numpy as np
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
from bench import load_package
load_package()
from cuda_ray_tracer_amd import binding, scenes
for cfg_no in (3, 2):
    spec, cfg, w, h = scenes.config_scene(cfg_no)
    built = scenes.build_scene(spec)
    r = binding.Renderer(0)
    r.upload(built.desc); r.set_state(built.camera, cfg)
    r.render(w, h)
    clk = r.profile_waves(w, h); clk = r.profile_waves(w, h)
    valid = clk[:, 1] > 0
    dur = (clk[valid, 1] - clk[valid, 0]).astype(np.float64)
    raw = clk[valid, 2:].astype(np.float64)
    both = raw[:, 12]; fusable = raw[:, 13]
    turns = (clk[valid, 2 + 15] & np.uint64(0xffffffff)).astype(np.float64)
    heavy = dur > np.percentile(dur, 99)
    for label, sel in (("all", np.ones(len(dur), bool)), ("heavy 1%", heavy)):
        print("config", cfg_no, label, "turns/wave %.1f, turns with both pools non-empty %.1f (%.0f%%), of those with both fitting one turn %.1f (%.0f%% of turns)" % (
            turns[sel].mean(), both[sel].mean(), 100 * both[sel].sum() / turns[sel].sum(), fusable[sel].mean(), 100 * fusable[sel].sum() / turns[sel].sum()))
