import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from bench import load_package
load_package()
import torch
from cuda_ray_tracer_amd import binding, scenes
spec, cfg, w, h = scenes.config_scene(3)
built = scenes.build_scene(spec)
r = binding.Renderer(0); r.upload(built.desc); r.set_state(built.camera, cfg)
frame = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render_device(w, h, frame.data_ptr(), stream)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-frame ms:", " ".join("%.3f" % t for t in ts))
clk = r.profile_waves(w, h)
regular = 32640 + 0
valid = clk[:, 1] > 0
dur = (clk[:, 1].astype(np.int64) - clk[:, 0].astype(np.int64)) / 100.0
t0 = clk[valid, 0].astype(np.int64).min()
ex = np.nonzero(valid[regular:])[0] + regular
rg = np.nonzero(valid[:regular])[0]
print("regular waves %d: max %.1f us; split waves %d (%d blocks): median %.1f max %.1f us; frame span %.1f us" % (
    len(rg), dur[rg].max(), len(ex), len(ex) // 8, np.median(dur[ex]) if len(ex) else 0, dur[ex].max() if len(ex) else 0,
    (clk[valid, 1].astype(np.int64).max() - t0) / 100.0))
print("split waves start (us) min/max:", (clk[ex, 0].astype(np.int64).min() - t0) / 100.0 if len(ex) else None, (clk[ex, 0].astype(np.int64).max() - t0) / 100.0 if len(ex) else None)
print("sum of wave time: regular %.1f ms, split %.1f ms" % (dur[rg].sum() / 1e3, dur[ex].sum() / 1e3))
