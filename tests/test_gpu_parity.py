"""GPU parity tests proper: the HIP path through the C ABI against the CPU oracle on identical inputs.

Bar: primary-hit ids identical on >= 99.99 % of pixels and RGB within 1 LSB (8-bit); ray counts equal.
Apart from libm calls (pow in the specular term; sin/cos/acos for soft-shadow samples) the two sides run the same
IEEE binary32 operations in the same order, so in practice ids and hit distances agree bit for bit.
"""
import numpy as np
import pytest

from conftest import load_package
from parity import assert_parity, compare_frames

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(renderer, oracle, spec, cfg, w, h, window=None, **build_kw):
    built = scenes.build_scene(spec, **build_kw)
    renderer.upload(built.desc)
    renderer.set_state(built.camera, cfg)
    g_rgb, g_ids, g_t, st = renderer.render_ids(w, h)
    plain_rgb, st2 = renderer.render(w, h)
    assert np.array_equal(plain_rgb, g_rgb), "instrumented and plain kernels disagree"
    assert st.rays == st2.rays
    o_rgb, o_ids, o_t, o_rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, window=window, threads=16)
    stats = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, window)
    return stats, st, o_rays


def test_config1_default_scene_640x480(renderer, oracle):
    spec, cfg, w, h = scenes.config_scene(1)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, w, h)
    print("config1", stats, st.rays)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


@pytest.mark.parametrize("key", ["Z", "X", "C", "V", "B", "N"])
def test_reference_analytic_scenes_nonsquare(renderer, oracle, key):
    index, cfg = scenes.preset_config(key)
    spec = scenes.reference_scene(index)
    scenes.apply_frame_fixups(spec, cfg)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, 320, 180)
    print(key, stats, st.rays)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


@pytest.mark.parametrize("key", ["M", ".", ","])
def test_reference_mesh_scenes(renderer, oracle, key):
    index, cfg = scenes.preset_config(key)
    spec = scenes.reference_scene(index, detail=0.5)
    scenes.apply_frame_fixups(spec, cfg)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, 256, 144)
    print(key, stats, st.rays, st.node_tests, st.tri_tests)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


def test_config2_cube_chalice_octree(renderer, oracle):
    spec, cfg, w, h = scenes.config_scene(2, detail=0.5)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, 320, 180)
    print("config2", stats, st.rays)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0


def test_config3_mesh_scene_reduced(renderer, oracle):
    spec, cfg, w, h = scenes.config_scene(3, detail=0.25)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, 384, 216)
    print("config3", stats, st.rays)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0


def test_config3_full_size_window(renderer, oracle):
    """Full 1080p frame and full-detail meshes on the GPU; the oracle checks a window the CPU finishes in seconds."""
    spec, cfg, w, h = scenes.config_scene(3)
    window = (900, 420, 1020, 480)
    stats, st, o_rays = _run(renderer, oracle, spec, cfg, w, h, window=window)
    print("config3 window", stats, st.rays)
    assert_parity(stats)
    assert stats["id_mismatch"] == 0


@pytest.mark.parametrize("leaf,depth", [(1, 12), (4, 8), (16, 3), (100000, 1)])
def test_octree_shape_does_not_change_pixels(renderer, oracle, leaf, depth):
    """Any octree is only an accelerator: frames must be identical for every leaf size / depth cap."""
    spec, cfg, w, h = scenes.config_scene(3, detail=0.2)
    built = scenes.build_scene(spec, octree_max_depth=depth, octree_leaf_triangles=leaf)
    renderer.upload(built.desc)
    renderer.set_state(built.camera, cfg)
    rgb, ids, t, st = renderer.render_ids(256, 144)
    built0 = scenes.build_scene(spec)
    renderer.upload(built0.desc)
    renderer.set_state(built0.camera, cfg)
    rgb0, ids0, t0, st0 = renderer.render_ids(256, 144)
    assert np.array_equal(rgb, rgb0) and np.array_equal(ids, ids0) and np.array_equal(t.view(np.uint32), t0.view(np.uint32))
    assert st.rays == st0.rays
