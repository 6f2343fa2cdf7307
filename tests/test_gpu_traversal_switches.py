"""The mesh traversal has two scheduling choices that may change the order of the work but never a pixel: a trace can
start at the model roots instead of the forest root (scenes with few models), and the hit children of a node can get
pool entries of their own while the pool holds fewer entries than the wave has lanes.  RT_TRAV (read when a context is
created) selects them: bit 0 = start at model roots, bit 1 = separate entries, bits 8-15 = entry limit / 8.  Every
combination must render the frame of the default setting byte for byte and agree with the oracle."""
import os

import numpy as np
import pytest

from conftest import load_package
from parity import assert_parity, compare_frames

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 2 | (8 << 8), 3 | (1 << 8), 3 | (63 << 8)])
def switched_renderer(request):
    old = os.environ.get("RT_TRAV")
    os.environ["RT_TRAV"] = str(request.param)
    r = binding.Renderer(0)
    yield r
    r.close()
    if old is None:
        os.environ.pop("RT_TRAV", None)
    else:
        os.environ["RT_TRAV"] = old


@pytest.mark.parametrize("mode", ["default", "eightboxes", "nobox"])
def test_traversal_switches_render_the_same_frame(renderer, switched_renderer, oracle, mode):
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.25)
    if mode == "eightboxes":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 1
    elif mode == "nobox":
        cfg.boundingBox = 0; cfg.eightBoundingBoxes = 0
    w, h = 320, 180
    built = scenes.build_scene(spec)
    for r in (renderer, switched_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    rgb, st = switched_renderer.render(w, h)
    assert np.array_equal(rgb, plain)
    assert st.rays == st0.rays
    g_rgb, g_ids, g_t, st = switched_renderer.render_ids(w, h)
    o_rgb, o_ids, o_t, o_rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=16)
    stats = compare_frames(g_rgb, g_ids, g_t, o_rgb, o_ids, o_t, None)
    assert_parity(stats, st.rays, o_rays)
    assert stats["id_mismatch"] == 0 and stats["t_bit_mismatch"] == 0


def test_traversal_switches_with_many_models(renderer, switched_renderer):
    """More models than a trace may start at: the switch for model roots must fall back to the forest root."""
    spec, cfg, _, _ = scenes.config_scene(4, detail=0.3)
    built = scenes.build_scene(spec)
    assert built.desc.modelCount > 6
    w, h = 480, 270
    for r in (renderer, switched_renderer):
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
    plain, st0 = renderer.render(w, h)
    rgb, st = switched_renderer.render(w, h)
    assert np.array_equal(rgb, plain) and st.rays == st0.rays
