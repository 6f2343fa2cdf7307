"""Golden frames (tests/golden/*.npz, written by tests/golden/make_golden.py with the CPU oracle).

CPU: the oracle still reproduces them exactly.  GPU: the HIP path reproduces them to the parity bar (RGB within
1 LSB, primary-hit ids identical; in practice bit for bit) without the oracle being run."""
import os

import numpy as np
import pytest

from conftest import ROOT, load_package
from parity import assert_parity, compare_frames

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402

import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

CASES = make_golden.cases()


def _load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return z["rgb"], z["ids"], z["t"], int(z["rays"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle, name):
    spec, cfg, w, h = CASES[name]
    built = scenes.build_scene(spec)
    rgb, ids, t, rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=8)
    g_rgb, g_ids, g_t, g_rays = _load(name)
    assert np.array_equal(rgb, g_rgb) and np.array_equal(ids, g_ids) and np.array_equal(t.view(np.uint32), g_t.view(np.uint32))
    assert rays == g_rays


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_reproduces_golden(renderer, name):
    spec, cfg, w, h = CASES[name]
    built = scenes.build_scene(spec)
    renderer.upload(built.desc)
    renderer.set_state(built.camera, cfg)
    rgb, ids, t, st = renderer.render_ids(w, h)
    g_rgb, g_ids, g_t, g_rays = _load(name)
    stats = compare_frames(rgb, ids, t, g_rgb, g_ids, g_t)
    assert_parity(stats, st.rays, g_rays)
    assert stats["id_mismatch"] == 0
