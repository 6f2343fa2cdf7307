"""Known-answer tests for the CPU oracle: values worked out by hand from the reference's formulas, plus the
reference quirks that change pixels.  (The reference ships no golden vectors for this path.  tests/test_ref_pins.py
checks the routines that compile with g++ against the reference's own compiled code; the tests here pin the rest - boxes,
shading, the trace loops - against the reference's source text.)"""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import load_package
from helpers import MATTE, mesh_from_triangles, simple_spec

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

f32 = np.float32


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def test_solve_quadratic(oracle):
    x0, x1 = C.c_float(), C.c_float()
    assert oracle.lib.oracle_solve_quadratic(1, -3, 2, x0, x1) == 1 and (x0.value, x1.value) == (1.0, 2.0)
    assert oracle.lib.oracle_solve_quadratic(1, 2, 1, x0, x1) == 1 and (x0.value, x1.value) == (-1.0, -1.0)
    assert oracle.lib.oracle_solve_quadratic(1, 0, 1, x0, x1) == 0
    # b > 0 branch: q = -0.5 (b + sqrt(disc))
    assert oracle.lib.oracle_solve_quadratic(1, 3, 2, x0, x1) == 1 and (x0.value, x1.value) == (-2.0, -1.0)


def test_wang_hash_matches_integer_definition(oracle):
    def ref(seed):
        m = 0xFFFFFFFF
        seed = ((seed ^ 61) ^ (seed >> 16)) & m
        seed = (seed * 9) & m
        seed = (seed ^ (seed >> 4)) & m
        seed = (seed * 0x27d4eb2d) & m
        seed = (seed ^ (seed >> 15)) & m
        return f32(f32(seed) / f32(4294967296.0))
    for s in [0, 1, 61, 12345, 0xFFFFFFFF, 1973 * 1000, 2 ** 31]:
        assert f32(oracle.lib.oracle_wang_hash(s)) == ref(s)


def test_sphere_hit_cases(oracle):
    t0, t1 = C.c_float(), C.c_float()
    s = T.Sphere(T.Vec3(0, 0, -10), 2.0, T.material(*MATTE), 0, 0)
    assert oracle.lib.oracle_sphere_hit(s, _f3((0, 0, 0)), _f3((0, 0, -1)), t0, t1) == 1 and (t0.value, t1.value) == (8.0, 12.0)
    # origin inside: the negative root is replaced by the positive one
    assert oracle.lib.oracle_sphere_hit(s, _f3((0, 0, -10)), _f3((0, 0, -1)), t0, t1) == 1 and (t0.value, t1.value) == (2.0, 2.0)
    # sphere behind the ray
    assert oracle.lib.oracle_sphere_hit(s, _f3((0, 0, -20)), _f3((0, 0, -1)), t0, t1) == 0
    # miss
    assert oracle.lib.oracle_sphere_hit(s, _f3((5, 0, 0)), _f3((0, 0, -1)), t0, t1) == 0


def test_plane_hit_cases(oracle):
    t0 = C.c_float()
    p = T.Plane(T.Vec3(0, 20, 0), T.Vec3(0, 1, 0), T.material(*MATTE), 0, 0)
    assert oracle.lib.oracle_plane_hit(p, _f3((0, 0, 0)), _f3((0, 1, 0)), t0) == 1 and t0.value == 20.0
    assert oracle.lib.oracle_plane_hit(p, _f3((0, 0, 0)), _f3((0, -1, 0)), t0) == 0 and t0.value == -20.0
    # parallel ray: 20 / 0 = +inf, which passes t >= 0 (it can never be the closest hit)
    assert oracle.lib.oracle_plane_hit(p, _f3((0, 0, 0)), _f3((1, 0, 0)), t0) == 1 and math.isinf(t0.value)


def test_aabb_hit_and_normal(oracle):
    t0, t1 = C.c_float(), C.c_float()
    lo, hi = _f3((-1, -1, -11)), _f3((1, 1, -9))
    assert oracle.lib.oracle_aabb_hit(lo, hi, _f3((0, 0, 0)), _f3((0, 0, -1)), t0, t1) == 1 and (t0.value, t1.value) == (9.0, 11.0)
    # origin inside: tmin < 0 is replaced by tmax
    lo, hi = _f3((-1, -1, -1)), _f3((1, 1, 1))
    assert oracle.lib.oracle_aabb_hit(lo, hi, _f3((0, 0, 0)), _f3((0, 0, -1)), t0, t1) == 1 and (t0.value, t1.value) == (1.0, 1.0)
    # behind
    assert oracle.lib.oracle_aabb_hit(lo, hi, _f3((0, 0, -5)), _f3((0, 0, -1)), t0, t1) == 0
    # 0/0 on an axis-parallel ray starting on a slab plane gives NaN, which the comparisons let through
    # (tmax_x = 0/0 stays NaN through every comparison, so t1 comes out NaN while t0 is the z-slab entry)
    assert oracle.lib.oracle_aabb_hit(lo, hi, _f3((1, 0, 5)), _f3((0, 0, -1)), t0, t1) == 1 and t0.value == 4.0 and math.isnan(t1.value)
    n = (C.c_float * 3)()
    oracle.lib.oracle_aabb_normal_at(lo, hi, _f3((1, 0.2, 0.3)), n)
    assert tuple(n) == (1.0, 0.0, 0.0)
    oracle.lib.oracle_aabb_normal_at(lo, hi, _f3((1, 1, 0)), n)
    assert n[0] == pytest.approx(1 / math.sqrt(2)) and n[1] == pytest.approx(1 / math.sqrt(2)) and n[2] == 0.0
    oracle.lib.oracle_aabb_normal_at(lo, hi, _f3((0.5, 0.5, 0.5)), n)     # strictly inside: (0,0,0)/0 -> NaN
    assert all(math.isnan(v) for v in n)


def test_triangle_intersect_and_cull_threshold(oracle):
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    p0, p1, p2 = _f3((0, 0, -5)), _f3((1, 0, -5)), _f3((0, 1, -5))
    o, d = _f3((0.25, 0.25, 0)), _f3((0, 0, -1))
    # det = +1: rejected when culling (only det < -0.01 survives), accepted otherwise
    assert oracle.lib.oracle_triangle_intersect(p0, p1, p2, o, d, 1, t, u, v) == 0
    assert oracle.lib.oracle_triangle_intersect(p0, p1, p2, o, d, 0, t, u, v) == 1
    assert (t.value, u.value, v.value) == (5.0, 0.25, 0.25)
    # reversed winding: det = -1 passes the cull
    assert oracle.lib.oracle_triangle_intersect(p0, p2, p1, o, d, 1, t, u, v) == 1 and t.value == 5.0
    # the threshold is the double -0.01: det = -float32(0.01) (> -0.01) is culled, the next float below is kept
    s_in = f32(0.01)
    s_out = np.nextafter(s_in, f32(1))
    for s, expect in ((s_in, 0), (s_out, 1)):
        q0, q1, q2 = _f3((0, 0, -5)), _f3((0, 1, -5)), _f3((float(s), 0, -5))     # det = -s exactly
        assert oracle.lib.oracle_triangle_intersect(q0, q1, q2, _f3((float(s) / 4, 0.25, 0)), d, 1, t, u, v) == expect
    # the standalone Triangle object uses |det| < 0.001 and no cull
    t0 = C.c_float()
    assert oracle.lib.oracle_triangle_object_hit(p0, p1, p2, o, d, t0) == 1 and t0.value == 5.0


def test_camera_ray_ignores_aspect_ratio(oracle, pkg):
    from cuda_ray_tracer_amd import binding
    cam = binding.camera_look((0, 0, 0), 90.0, -90.0, 0.0)
    assert cam.fovFactor == pytest.approx(1.0, abs=1e-6)
    assert (cam.direction.z, cam.right.x, cam.up.y) == (-1.0, 1.0, 1.0)
    out = (C.c_float * 3)()
    # pixel centre (0.5, 0.5) of a 4x2 frame: sx = -0.75, sy = 0.5; x is NOT scaled by 4/2
    oracle.lib.oracle_camera_ray(cam, 0.5, 0.5, 4, 2, out)
    assert out[0] / -out[2] == pytest.approx(-0.75, rel=1e-6) and out[1] / -out[2] == pytest.approx(0.5, rel=1e-6)
    assert math.sqrt(sum(c * c for c in out)) == pytest.approx(1.0, rel=1e-6)


def test_pack_channel(oracle):
    pc = oracle.lib.oracle_pack_channel
    assert pc(0.0) == 0 and pc(1.0) == 255 and pc(2.5) == 255 and pc(0.5) == 127 and pc(-0.3) == 0
    assert pc(float("nan")) == 255          # min(NaN, 1) selects 1
    assert pc(0.999) == 254                 # truncation, not rounding


def _render_px(oracle, spec, cfg, w, h, x, y):
    built = scenes.build_scene(spec)
    rgb, ids, t, rays, _ = oracle.render(built.desc, built.camera, cfg, w, h, window=(x, y, x + 1, y + 1), threads=1)
    return rgb[y, x], ids[y, x], t[y, x], rays


def test_far_clip_is_999(oracle):
    cfg = scenes.default_config()
    for z, visible in ((-990.0, True), (-1010.0, False)):
        spec = simple_spec(spheres=[((0, 0, z), 5.0, MATTE, False, 0)])
        rgb, ids, t, rays = _render_px(oracle, spec, cfg, 3, 3, 1, 1)
        assert (ids[0] != -1) == visible
        assert (t < 999.0) == visible


def test_last_triangle_of_a_model_is_never_tested(oracle):
    cfg = scenes.default_config(); cfg.cullBackTriangles = 0
    a = [[-10, -10, -5], [10, -10, -5], [0, 10, -5]]
    b = [[-10, -10, -8], [10, -10, -8], [0, 10, -8]]
    one = mesh_from_triangles([a])           # a single triangle is also the last one: invisible
    two = mesh_from_triangles([a, b])        # only the first of two is reachable
    two_swapped = mesh_from_triangles([b, a])
    for mesh, expect_t in ((one, 999.0), (two, 5.0), (two_swapped, 8.0)):
        spec = simple_spec(models=[scenes.ModelSpec("test", (0, 0, 0), 1, MATTE)])
        built = scenes.build_scene(spec, meshes={"test": mesh})
        _, ids, t, _, _ = oracle.render(built.desc, built.camera, cfg, 3, 3, window=(1, 1, 2, 2), threads=1)
        assert t[1, 1] == f32(expect_t)


def test_barycentric_weights_are_permuted(oracle):
    """u multiplies the FIRST vertex's attributes (Model.cpp:300-310), although in Moeller-Trumbore u is the weight of
    the second: a hit next to vertex 1 is shaded with vertex 0's colour."""
    cfg = scenes.default_config(); cfg.cullBackTriangles = 0; cfg.diffuseLighting = 0; cfg.specularLighting = 0
    tri = [[-4, -4, -5], [4, -4, -5], [-4, 4, -5]]
    red, green, blue = [1, 0, 0], [0, 1, 0], [0, 0, 1]
    mesh = mesh_from_triangles([tri, tri], ka=[[red, green, blue], [red, green, blue]])
    mat = ((0, 0, 0), (0, 0, 0), 1.0, 0.0, 0.0, 1.0)        # ambient only: colour = interpolated Ka
    spec = simple_spec(models=[scenes.ModelSpec("test", (0, 0, 0), 1, mat)])
    built = scenes.build_scene(spec, meshes={"test": mesh})
    w = h = 101
    # world y is mirrored (Model.cpp:93): the vertices sit at world (-4, 4), (4, 4), (-4, -4) on the plane z = -5
    rgb, ids, t, _, _ = oracle.render(built.desc, built.camera, cfg, w, h)
    ys, xs = np.nonzero(ids[..., 0] != -1)
    assert len(xs) > 100
    px = lambda wx, wy: (int(round((wx / 5.0 + 1) * w / 2 - 0.5)), int(round((1 - wy / 5.0) * h / 2 - 0.5)))
    x1, y1 = px(3.2, 3.6)       # next to vertex 1: Moeller-Trumbore u ~ 0.9
    assert rgb[y1, x1, 0] > 200 and rgb[y1, x1, 1] < 40     # red = vertex 0's colour
    x0, y0 = px(-3.6, 3.6)      # next to vertex 0: u ~ 0, v ~ 0, so the third vertex's colour
    assert rgb[y0, x0, 2] > 200 and rgb[y0, x0, 0] < 40
    x2, y2 = px(-3.6, -3.2)     # next to vertex 2: v ~ 0.9, so the second vertex's colour
    assert rgb[y2, x2, 1] > 200 and rgb[y2, x2, 2] < 40


def test_reflection_depth_quirk(oracle):
    """maxDepth 1 -> one reflection cast; any maxDepth >= 2 -> three chained casts (kernels.cu:168-170, 198-200)."""
    index, cfg = scenes.preset_config("N")
    spec = scenes.reference_scene(index)
    scenes.apply_frame_fixups(spec, cfg)
    built = scenes.build_scene(spec)
    counts = {}
    for depth in (1, 2, 3, 50):
        cfg.maxDepth = depth
        _, _, _, rays, _ = oracle.render(built.desc, built.camera, cfg, 160, 90, threads=8, want_ids=False)
        counts[depth] = rays
    assert counts[2] == counts[3] == counts[50] > counts[1] > 160 * 90
    assert counts[1] <= 2 * 160 * 90 and counts[2] <= 4 * 160 * 90


def test_octree_flag_hides_meshes_in_the_reference(oracle):
    """Model::hit returns false for every ray when useOctree is set (Model.cpp:163-219)."""
    spec, cfg, _, _ = scenes.config_scene(2, detail=0.2)
    built = scenes.build_scene(spec)
    cfg.octree = 1
    _, ids, _, _, _ = oracle.render(built.desc, built.camera, cfg, 96, 54, threads=8)
    assert not np.any((ids[..., 0] >> 24) == T.RT_MODEL_T)
    cfg.octree = 0
    _, ids, _, _, _ = oracle.render(built.desc, built.camera, cfg, 96, 54, threads=8)
    assert np.any((ids[..., 0] >> 24) == T.RT_MODEL_T)


def test_eight_box_and_no_box_modes_run(oracle):
    spec, cfg, _, _ = scenes.config_scene(2, detail=0.2)
    built = scenes.build_scene(spec)
    base, _, _, _, _ = oracle.render(built.desc, built.camera, cfg, 96, 54, threads=8)
    cfg.eightBoundingBoxes = 1
    eight, _, _, _, _ = oracle.render(built.desc, built.camera, cfg, 96, 54, threads=8)
    cfg.eightBoundingBoxes = 0; cfg.boundingBox = 0
    none, _, _, _, _ = oracle.render(built.desc, built.camera, cfg, 96, 54, threads=8)
    # the modes differ only where triangles are dropped per chunk box or, without a box, where a hit with t <= 0 counts
    for other in (eight, none):
        frac = float((np.abs(other.astype(int) - base.astype(int)).max(axis=-1) > 1).mean())
        assert frac < 0.15, frac
