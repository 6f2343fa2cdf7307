"""The mesh traversal keeps the pending work of a wave's 64 rays in two LDS pools.  When the pool of pending children
has fewer than 88 free entries, single lanes descend depth-first until there is room again; when the pool of triangle
work (runs of triangles and sets of leaves) has fewer than 64 free entries, a triangle turn takes fewer entries and at
the limit a single lane works on the newest one.  The compiled capacities (512 and 288) are rarely exhausted, so these
tests shrink them (RT_NODE_POOL, RT_TRI_POOL, read when a context is created) until those paths carry most of the
traversal, and require byte-identical frames and oracle parity."""
import os

import numpy as np
import pytest

from conftest import load_package
from parity import assert_parity, compare_frames

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(96, 0), (160, 72), (0, 72), (0, 100)])
def small_pool_renderer(request):
    old = {k: os.environ.get(k) for k in ("RT_NODE_POOL", "RT_TRI_POOL")}
    os.environ["RT_NODE_POOL"], os.environ["RT_TRI_POOL"] = str(request.param[0]), str(request.param[1])
    r = binding.Renderer(0)
    yield r
    r.close()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("mode", ["default", "eightboxes", "nobox", "deep"])
def test_small_node_pool_renders_the_same_frame(renderer, small_pool_renderer, oracle, mode):
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.25)
    kw = {}
    if mode == "eightboxes":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 1
    elif mode == "nobox":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 0
    elif mode == "deep":
        kw = dict(octree_max_depth=10, octree_leaf_triangles=1)     # tall trees: many entries per ray
    w, h = 320, 180
    built = scenes.build_scene(spec, **kw)
    for r in (renderer, small_pool_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    rgb, st = small_pool_renderer.render(w, h)
    assert np.array_equal(rgb, plain)
    assert st.rays == st0.rays
    g_rgb, g_ids, g_t, st = small_pool_renderer.render_ids(w, h)
    o_rgb, o_ids, o_t, o_rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=16)
    stats = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, None)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


def test_full_size_frame_with_small_pool_matches(renderer, small_pool_renderer):
    """The 1080p benchmark frame, full-detail meshes: a small pool changes the order of the work, never a pixel."""
    spec, cfg, w, h = scenes.config_scene(3)
    built = scenes.build_scene(spec)
    for r in (renderer, small_pool_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    rgb, st = small_pool_renderer.render(w, h)
    assert np.array_equal(rgb, plain)
    assert st.rays == st0.rays


@pytest.mark.parametrize("mode", ["whole", "eightboxes"])
def test_many_models_gates_on_the_fly(renderer, small_pool_renderer, oracle, mode):
    """More than 64 models: no up-front gate pass, the lane that takes a model-root entry evaluates that model's gates
    (and in the eight-box mode its chunk gates) for the entry's ray."""
    spec, cfg, _, _ = scenes.config_scene(5, detail=0.15)
    if mode == "eightboxes":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 1
    w, h = 320, 180
    built = scenes.build_scene(spec)
    assert built.desc.modelCount > 64
    for r in (renderer, small_pool_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    rgb, st = small_pool_renderer.render(w, h)
    assert np.array_equal(rgb, plain) and st.rays == st0.rays
    window = (120, 60, 200, 100)
    g_rgb, g_ids, g_t, st = renderer.render_ids(w, h)
    o_rgb, o_ids, o_t, _, _ = oracle.render(built.desc, built.camera, cfg, w, h, window=window, threads=16)
    stats = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, window)
    assert_parity(stats)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0
