"""The CPU oracle against the reference-generated fixtures (tests/golden/refgen_*.npz): primary-hit ids and distances of
whole frames, ray directions, and Model::hit without a box.  These run without the reference checkout; where oracle/_ref
exists, the committed fixtures are also regenerated in memory and must come out byte for byte."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import ref_golden
import ref_lib
from conftest import ROOT, load_package

load_package()
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

pytestmark = pytest.mark.filterwarnings("ignore::RuntimeWarning")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_fixtures_exist():
    assert len(ref_golden.analytic_names()) == 6 and len(ref_golden.mesh_names()) == 3


@pytest.mark.parametrize("name", ref_golden.analytic_names())
def test_oracle_reproduces_the_reference_frames(oracle, name):
    fx = ref_golden.load(name)
    built, cfg, w, h = ref_golden.analytic_scene(fx)
    _, ids, t, _, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=8)
    ref_golden.assert_analytic_frame(ids, t, fx, "oracle")


def test_oracle_reproduces_the_reference_rays(oracle):
    fx = ref_golden.load("refgen_rays")
    for i, (cam, (w, h)) in enumerate(zip(fx["cams"], fx["sizes"])):
        basis = oracle.camera_basis(cam[0:3], float(cam[3]), float(cam[4]), float(cam[5]))
        got = np.float32([*basis.position.tuple(), *basis.up.tuple(), *basis.right.tuple(), *basis.direction.tuple(), basis.fovFactor])
        assert np.array_equal(got.view(np.uint32), fx[f"basis{i}"].view(np.uint32))
        for j, off in enumerate(fx["offsets"]):
            raw = np.zeros((h, w, 3), np.float32); direction = np.zeros((h, w, 3), np.float32)
            oracle.lib.oracle_kat_camera_rays(C.byref(basis), int(w), int(h), 0, 0, int(w), int(h), C.c_double(off[0]), C.c_double(off[1]), _p(raw), _p(direction))
            assert np.array_equal(raw.view(np.uint32), fx[f"raw{i}_{j}"].view(np.uint32))
            assert np.array_equal(direction.view(np.uint32), fx[f"dir{i}_{j}"].view(np.uint32))


@pytest.mark.parametrize("name", ref_golden.mesh_names())
def test_oracle_reproduces_the_reference_mesh_hits(oracle, name):
    fx = ref_golden.load(name)
    built, cfg, w, h = ref_golden.mesh_scene(fx)
    rgb, ids, t, _, _ = oracle.render(built.desc, built.camera, cfg, w, h, threads=8)
    ref_golden.assert_mesh_frame(ids, t, fx, "oracle")
    hit = fx["hit"] == 1
    assert np.array_equal(rgb[hit], ref_golden.ambient_colour_bytes(fx)[hit])            # the interpolated Ka reaches the pixel
    # and the routine by itself: interpolated normal / Ka / Kd
    verts12, indices = ref_golden.model_arrays(built.desc.models[0])
    n = w * h
    p = _p
    cam = fx["cam_pos"]
    raw = np.zeros((h, w, 3), np.float32); direction = np.zeros((h, w, 3), np.float32)
    oracle.lib.oracle_kat_camera_rays(C.byref(built.camera), w, h, 0, 0, w, h, C.c_double(0.5), C.c_double(0.5), p(raw), p(direction))
    rays = np.concatenate([np.tile(np.float32(cam), (n, 1)), direction.reshape(-1, 3)], axis=1).astype(np.float32)
    o_hit = np.zeros(n, np.int32); o_t = np.zeros(n, np.float32); o_attrs = np.zeros((n, 9), np.float32)
    oracle.lib.oracle_kat_model_hit_nobox(len(verts12), p(verts12), len(indices), p(indices), n, p(rays), int(fx["cull"]), p(o_hit), p(o_t), p(o_attrs))
    assert np.array_equal(o_hit.reshape(h, w) == 1, hit)
    assert np.array_equal(o_attrs.reshape(h, w, 9).view(np.uint32), fx["attrs"].view(np.uint32))
    if name == "refgen_mesh_cube":
        assert (fx["t0"][hit] < 0).sum() > 1000          # camera inside the cube: without a box the nearest hit may lie behind it


@pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref is not built and the reference checkout is absent")
def test_committed_fixtures_are_what_the_reference_produces_now():
    spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(ROOT, "tests", "golden", "make_ref_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    seen = 0
    for name, arrays in gen.fixtures(ref_lib.ref()):
        stored = np.load(os.path.join(ref_golden.GOLDEN, name))
        assert sorted(stored.files) == sorted(arrays)
        for key in arrays:
            a, b = np.asarray(arrays[key]), stored[key]
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (name, key)
        seen += 1
    assert seen == 10
