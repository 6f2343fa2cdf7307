"""Split waves: the heaviest 8x8 blocks of a frame are rendered in the next frame by eight waves, one per pixel row (eight
lanes bring a pixel, the other 56 only help with the mesh traversal).  RT_SPLIT_RATIO=0 makes the selection take every
block that fits, so that most pixels of these small frames go through that path; the frames must not change by a single
byte, and hit ids / distances must match the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import load_package
from parity import assert_parity, compare_frames

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_renderer():
    old = {k: os.environ.get(k) for k in ("RT_SPLIT_RATIO", "RT_SPLIT_FLOOR")}
    os.environ["RT_SPLIT_RATIO"] = "0"
    os.environ["RT_SPLIT_FLOOR"] = "0"
    r = binding.Renderer(0)
    yield r
    r.close()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("mode", ["default", "antialias+soft", "eightboxes", "nobox"])
def test_split_frames_equal_unsplit_frames(renderer, split_renderer, oracle, mode):
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.25)
    if mode == "antialias+soft":
        cfg.antiAliasing = 1; cfg.renderSoftShadows = 1; cfg.softShadowNum = 3
    elif mode == "eightboxes":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 1
    elif mode == "nobox":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 0
    w, h = 320, 180
    built = scenes.build_scene(spec)
    for r in (renderer, split_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    frames = [split_renderer.render(w, h) for _ in range(3)]       # frame 1 records costs, frames 2 and 3 are split
    for rgb, st in frames:
        assert np.array_equal(rgb, plain)
        assert st.rays == st0.rays
    # ids and hit distances through the split path (the instrumented kernel uses the same feedback)
    g_rgb, g_ids, g_t, st = split_renderer.render_ids(w, h)
    o_rgb, o_ids, o_t, o_rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=16)
    stats = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, None)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


def test_split_tile_sharded_frame(split_renderer):
    import torch
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.2)
    built = scenes.build_scene(spec)
    split_renderer.upload(built.desc)
    split_renderer.set_state(built.camera, cfg)
    w, h, world = 400, 230, 3
    direct, _ = split_renderer.render(w, h)
    per = binding.tile_buffer_bytes(w, h, world)
    gathered = torch.zeros(per * world, dtype=torch.uint8, device="cuda:0")
    frame = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        for _ in range(3):   # a context keeps feedback for one (frame size, rank, world): launches 2 and 3 are split
            gathered[rank * per:(rank + 1) * per].zero_()
            split_renderer.render_tiles_device(w, h, rank, world, gathered.data_ptr() + rank * per, per, stream)
    split_renderer.untile_device(w, h, world, gathered.data_ptr(), frame.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(direct, frame.cpu().numpy().reshape(h, w, 3))


def test_geometry_change_through_a_path_launch_forgets_the_split_list(split_renderer):
    """A context that splits blocks of one frame size, then renders a camera path at another size, then single frames at
    that size: the split list made for the first size must not be used for the second (it indexes other blocks)."""
    import torch
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.2)
    built = scenes.build_scene(spec)
    split_renderer.upload(built.desc)
    split_renderer.set_state(built.camera, cfg)
    for _ in range(3):
        big, _ = split_renderer.render(640, 360)          # blocks of the large frame are split from the second frame on
    w, h = 160, 96
    stream = torch.cuda.current_stream().cuda_stream
    frames = torch.zeros(2 * w * h * 3, dtype=torch.uint8, device="cuda:0")
    guard = torch.full((1 << 20,), 77, dtype=torch.uint8, device="cuda:0")      # allocated right behind: must stay untouched
    split_renderer.render_path_device(w, h, [built.camera, built.camera], frames.data_ptr(), stream)
    small = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        split_renderer.render_device(w, h, small.data_ptr(), stream)
    torch.cuda.synchronize()
    ref, _ = split_renderer.render(w, h)
    assert np.array_equal(small.cpu().numpy().reshape(h, w, 3), ref)
    assert np.array_equal(frames.cpu().numpy()[:w * h * 3].reshape(h, w, 3), ref)
    assert bool((guard == 77).all())
