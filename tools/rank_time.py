import os, sys, time
ROOT = os.environ.get("RT_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from bench import load_package
load_package()
import torch
from cuda_ray_tracer_amd import binding, scenes
cfgno = int(os.environ.get("RT_CFG", "3"))
spec, cfg, w, h = scenes.config_scene(cfgno)
built = scenes.build_scene(spec)
r = binding.Renderer(0); r.upload(built.desc); r.set_state(built.camera, cfg)
stream = torch.cuda.current_stream().cuda_stream
for world in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    nbytes = binding.tile_buffer_bytes(w, h, world)
    tiles = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    worst = 0
    for rank in range(world):
        for _ in range(20): r.render_tiles_device(w, h, rank, world, tiles.data_ptr(), nbytes, stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): r.render_tiles_device(w, h, rank, world, tiles.data_ptr(), nbytes, stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        worst = max(worst, dt)
    print(f"{ROOT} cfg {cfgno} tile {binding.RT_TILE} world {world}: slowest rank {worst * 1e3:.3f} ms per frame (render only, no exchange)", flush=True)
