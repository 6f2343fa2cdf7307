#!/usr/bin/env python3
"""Prints the distribution of per-wave run times and the occupancy timeline of one frame (diagnostic)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

cfg_no = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spec, cfg, w, h = scenes.config_scene(cfg_no)
built = scenes.build_scene(spec)
r = binding.Renderer(0)
r.upload(built.desc); r.set_state(built.camera, cfg)
r.render(w, h)
clk = r.profile_waves(w, h)
clk = r.profile_waves(w, h)
valid = clk[:, 1] > 0
s, e = clk[valid, 0].astype(np.int64), clk[valid, 1].astype(np.int64)
t0 = s.min()
dur = (e - s) / 100.0          # microseconds
print(f"waves {valid.sum()}  frame span {(e.max() - t0) / 100.0:.1f} us  sum of wave time {dur.sum() / 1e3:.1f} ms  mean {dur.mean():.1f} us")
print("duration percentiles (us):", {p: round(float(np.percentile(dur, p)), 1) for p in (50, 90, 99, 99.9, 100)})
span = (e.max() - t0)
bins = 20
edges = np.linspace(0, span, bins + 1)
for i in range(bins):
    a, b = edges[i], edges[i + 1]
    active = (np.minimum(e - t0, b) - np.maximum(s - t0, a)).clip(min=0).sum() / (b - a)
    print(f"  {a / 100:8.1f}-{b / 100:8.1f} us: {active:7.1f} waves resident on average")
raw = clk[valid, 2:]
lanes = (raw[:, 1::2] >> np.uint64(32)).astype(np.float64)      # lanes active, summed over the executions of a section
sec = raw.astype(np.float64)
sec[:, 1::2] = (raw[:, 1::2] & np.uint64(0xffffffff)).astype(np.float64)
tot_cycles = dur * 2400.0   # rough: 2.4 GHz
print("share of wave time (all waves): node visits %.1f%%, triangle tests %.1f%%, model gates %.1f%%, whole mesh traces %.1f%%" % tuple(100 * sec[:, k].sum() / tot_cycles.sum() for k in (0, 2, 4, 6)))
heavy = dur > np.percentile(dur, 99)
print("heavy 1%% waves: node visits %.1f%% (%.0f iters/wave, %.0f cyc/iter), triangle tests %.1f%% (%.0f iters, %.0f cyc/iter), gates %.1f%% (%.0f iters, %.0f cyc/iter)" % (
    100 * sec[heavy, 0].sum() / tot_cycles[heavy].sum(), sec[heavy, 1].mean(), sec[heavy, 0].sum() / max(sec[heavy, 1].sum(), 1),
    100 * sec[heavy, 2].sum() / tot_cycles[heavy].sum(), sec[heavy, 3].mean(), sec[heavy, 2].sum() / max(sec[heavy, 3].sum(), 1),
    100 * sec[heavy, 4].sum() / tot_cycles[heavy].sum(), sec[heavy, 5].mean(), sec[heavy, 4].sum() / max(sec[heavy, 5].sum(), 1)))
order = np.argsort(-dur)[:10]
idx = np.nonzero(valid)[0][order]
names = ["node visits", "triangle tests", "model gates", "whole mesh traces", "linear-table traces", "surface + lighting", "linear-table shadows"]
for label, sel in (("all waves", np.ones(len(dur), bool)), ("heavy 1% waves", heavy)):
    print(label + ": " + ", ".join("%s %.1f%% (%.0f/wave, %.0f cyc each)" % (names[k], 100 * sec[sel, 2 * k].sum() / tot_cycles[sel].sum(), sec[sel, 2 * k + 1].mean(),
                                                                            sec[sel, 2 * k].sum() / max(sec[sel, 2 * k + 1].sum(), 1)) for k in range(7)))
for label, sel in (("all waves", np.ones(len(dur), bool)), ("heavy 1% waves", heavy)):
    print(label + ", lanes active per execution: " + ", ".join("%s %.1f" % (names[k], lanes[sel, k].sum() / max(sec[sel, 2 * k + 1].sum(), 1)) for k in range(7)))
for label, sel in (("all waves", np.ones(len(dur), bool)), ("heavy 1% waves", heavy)):
    turns = sec[sel, 15].sum()
    print(label + ": traversal turns %.0f/wave, lanes (pool entries) with work %.1f, lanes inside the trace call %.1f" % (sec[sel, 15].mean(), raw[sel, 14].astype(np.float64).sum() / max(turns, 1), lanes[sel, 7].sum() / max(turns, 1)))
print("heavy 1%% waves: whole mesh traces %.1f%% of their time (%.0f calls/wave, %.0f cyc/call)" % (
    100 * sec[heavy, 6].sum() / tot_cycles[heavy].sum(), sec[heavy, 7].mean(), sec[heavy, 6].sum() / max(sec[heavy, 7].sum(), 1)))
allidx = np.nonzero(valid)[0]
extra = allidx >= (len(clk) - 4096)
if extra.any():
    print("split waves: %d, duration percentiles (us): %s, sum %.1f ms" % (extra.sum(), {p: round(float(np.percentile(dur[extra], p)), 1) for p in (50, 90, 99, 100)}, dur[extra].sum() / 1e3))
    print("regular waves: duration percentiles (us): %s" % ({p: round(float(np.percentile(dur[~extra], p)), 1) for p in (50, 90, 99, 99.9, 100)},))
    slow = extra & (dur > np.percentile(dur[extra], 90))
    for label, sel in (("split waves", extra), ("slowest 10% of the split waves", slow)):
        turns = sec[sel, 15].sum()
        print(label + ": %.0f traversal turns/wave with %.1f entries each; node turns %.0f/wave (%.0f cyc), triangle turns %.0f/wave (%.0f cyc), mesh traces %.0f/wave, %.1f%% of the time in mesh traces" % (
            sec[sel, 15].mean(), raw[sel, 14].astype(np.float64).sum() / max(turns, 1), sec[sel, 1].mean(), sec[sel, 0].sum() / max(sec[sel, 1].sum(), 1),
            sec[sel, 3].mean(), sec[sel, 2].sum() / max(sec[sel, 3].sum(), 1), sec[sel, 7].mean(), 100 * sec[sel, 6].sum() / tot_cycles[sel].sum()))
print("slowest waves (wave index, start us, dur us):", [(int(i), round(float((clk[i, 0] - t0) / 100.0), 1), round(float((clk[i, 1] - clk[i, 0]) / 100.0), 1)) for i in idx])
