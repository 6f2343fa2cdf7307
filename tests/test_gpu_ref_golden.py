"""The HIP kernels, through the C ABI, against THE REFERENCE ITSELF — no oracle in between.

The expected values in tests/golden/refgen_*.npz were computed by the reference's own compiled Camera, Sphere, Plane and
Model routines (tests/golden/make_ref_golden.py, oracle/_ref).  Bar: primary-hit identities equal and hit distances equal
bit for bit on every pixel; where a mesh is hit under ambient-only lighting the pixel holds the reference's interpolated Ka.
"""
import numpy as np
import pytest

import ref_golden

pytestmark = pytest.mark.gpu


def _render(renderer, built, cfg, w, h):
    renderer.upload(built.desc)
    renderer.set_state(built.camera, cfg)
    rgb, ids, t, st = renderer.render_ids(w, h)
    plain, st2 = renderer.render(w, h)
    assert np.array_equal(plain, rgb) and st.rays == st2.rays, "instrumented and plain kernels disagree"
    return rgb, ids, t


@pytest.mark.parametrize("name", ref_golden.analytic_names())
def test_kernel_reproduces_the_reference_frames(renderer, name):
    """Camera::rasterToCameraSpace + normalise + Sphere::hit / Plane::hit + the closest-hit rule, reference scenes 1, 2 and 5
    at 640x480 and on non-square frames."""
    fx = ref_golden.load(name)
    built, cfg, w, h = ref_golden.analytic_scene(fx)
    _, ids, t = _render(renderer, built, cfg, w, h)
    ref_golden.assert_analytic_frame(ids, t, fx, "kernel")


@pytest.mark.parametrize("name", ref_golden.mesh_names())
@pytest.mark.parametrize("on_device", [False, True])
def test_kernel_reproduces_the_reference_mesh_hits(renderer, name, on_device):
    """Model::triangleIntersect + Model::hit without a box: the octree traversal must find the reference's brute-force
    closest hit (negative distances included) on host-built and GPU-built trees."""
    fx = ref_golden.load(name)
    built, cfg, w, h = ref_golden.mesh_scene(fx)
    built.desc.buildOnDevice = 1 if on_device else 0
    rgb, ids, t = _render(renderer, built, cfg, w, h)
    ref_golden.assert_mesh_frame(ids, t, fx, "kernel")
    hit = fx["hit"] == 1
    diff = np.abs(rgb[hit].astype(int) - ref_golden.ambient_colour_bytes(fx)[hit].astype(int))
    assert diff.max() == 0, f"interpolated Ka differs from the reference's on {(diff.max(axis=-1) > 0).sum()} pixels"
