"""The CPU oracle (and the product's host-side camera routine) against THE REFERENCE ITSELF: oracle/_ref holds the reference's
own utils.cpp, Sphere.cpp, Plane.cpp, Triangle.cpp, Camera.cpp and Model.cpp compiled in this image (oracle/build_ref.sh, no
stand-in headers, no source patch), and every comparison here is bit for bit on the same arrays.

What this pins: solveQuadratic (utils.cpp:10-32), Sphere::hit / normalAt (Sphere.cpp:17-48), Plane::hit / normalAt
(Plane.cpp:16-26), Triangle::hit (Triangle.cpp:20-42), the vec3 helpers (vec3.h:26, 102-117, utils.h:102-104), the camera
basis and ray generation (Camera.cpp:5-63 as called in kernels.cu:337-338, 355-356), Camera::move / mouseMovement
(Camera.cpp:40-52, 65-83), and Model::triangleIntersect with Model::hit's branch without a box and its attribute
interpolation (Model.cpp:135-159, 318-356).  What it cannot reach: kernels.cu and AABB.cpp do not compile with g++.
"""
import ctypes as C

import numpy as np
import pytest

import ref_lib
from conftest import load_package

load_package()
from cuda_ray_tracer_amd import binding, meshgen, scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

pytestmark = [pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref is not built and the reference checkout is absent"),
              pytest.mark.filterwarnings("ignore::RuntimeWarning")]

f32 = np.float32
N = 1_000_000


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same_floats(got, want, what, where=None):
    """Bit equality; `where` restricts the comparison to the cases the routine reports as hits."""
    g, w = bits(got), bits(want)
    if where is not None:
        g, w = g[where], w[where]
    bad = np.nonzero(g != w)
    assert bad[0].size == 0, f"{what}: {bad[0].size} of {g.size} differ, first at {tuple(b[0] for b in bad)}: {g[bad][0]:#x} vs {w[bad][0]:#x}"


@pytest.fixture(scope="module")
def ref():
    return ref_lib.ref()


def _vectors(rng, n, scale):
    """Random vectors over several magnitudes, with exact zeros sprinkled in."""
    v = rng.normal(size=(n, 3)) * scale * (10.0 ** rng.integers(-2, 3, size=(n, 1)))
    v[rng.random((n, 3)) < 0.02] = 0.0
    return f32(v)


def test_type_sizes_are_the_ones_the_layouts_assume(ref):
    assert ref.sizes() == {"vec3": 12, "Material": 40, "Sphere": 68, "Plane": 76, "Triangle": 100, "Camera": 84, "Vertex": 76,
                           "Light": 60, "Hit": 92}
    assert C.sizeof(T.Vertex) == 76 and C.sizeof(T.Material) == 40 and C.sizeof(T.Vec3) == 12     # same layouts across the C ABI


def test_solve_quadratic(ref, oracle):
    rng = np.random.default_rng(11)
    a = f32(rng.normal(size=N) * 10.0 ** rng.integers(-3, 4, N)); b = f32(rng.normal(size=N) * 10.0 ** rng.integers(-3, 4, N))
    c = f32(rng.normal(size=N) * 10.0 ** rng.integers(-3, 4, N))
    # forced discriminant == 0 (b*b and 4ac exact in float): a = m^2, c = n^2, b = +-2mn
    m = rng.integers(1, 200, 20000).astype(np.float32); k = rng.integers(1, 200, 20000).astype(np.float32)
    sgn = np.where(rng.random(20000) < 0.5, -1, 1).astype(np.float32)
    a[:20000], b[:20000], c[:20000] = m * m, sgn * 2 * m * k, k * k
    # both roots negative, a = 0, b = 0, zeros, infinities and NaN
    a[20000:30000] = 1; b[20000:30000] = np.abs(b[20000:30000]) + 3; c[20000:30000] = 1
    a[30000:31000] = 0; b[31000:32000] = 0; c[32000:33000] = 0
    a[33000:33100] = np.inf; b[33100:33200] = -np.inf; c[33200:33300] = np.nan; a[33300:33400] = -0.0
    ok_r, x0_r, x1_r = ref.solve_quadratic(a, b, c)
    ok_o = np.zeros(N, np.int32); x0_o = np.zeros(N, np.float32); x1_o = np.zeros(N, np.float32)
    oracle.lib.oracle_kat_solve_quadratic(N, _p(a), _p(b), _p(c), _p(ok_o), _p(x0_o), _p(x1_o))
    assert np.array_equal(ok_r, ok_o)
    assert 0.2 < ok_r.mean() < 0.9 and np.all(ok_r[:20000] == 1)
    assert np.all(x0_r[:20000] == x1_r[:20000])                    # the double-precision branch of a zero discriminant was taken
    assert_same_floats(x0_o, x0_r, "x0"); assert_same_floats(x1_o, x1_r, "x1")


def test_sphere_hit(ref, oracle):
    rng = np.random.default_rng(12)
    centre = _vectors(rng, N, 30); radius = f32(np.abs(rng.normal(size=N)) * 10 + 0.01)
    origin = _vectors(rng, N, 30); d = _vectors(rng, N, 1)
    # aimed at the sphere (most random rays miss): direction = towards a point near the centre, half of them normalised
    aim = centre + f32(rng.normal(size=(N, 3))) * radius[:, None] * 0.7 - origin
    use = rng.random(N) < 0.7
    d[use] = aim[use]
    nrm = rng.random(N) < 0.5
    with np.errstate(all="ignore"):
        d[nrm] = (d[nrm] / np.sqrt((d[nrm].astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))[:, None]).astype(np.float32)
    # origin inside the sphere, sphere behind the ray
    inside = slice(0, 50000); origin[inside] = centre[inside] + f32(rng.normal(size=(50000, 3))) * radius[inside, None] * 0.3
    behind = slice(50000, 100000); d[behind] = -aim[behind]
    # tangent rays with a zero discriminant: centre (0, r, -k), ray from the origin along -z, integers
    tang = slice(100000, 110000)
    r = rng.integers(1, 50, 10000).astype(np.float32); k = rng.integers(1, 50, 10000).astype(np.float32)
    centre[tang] = np.stack([np.zeros(10000, np.float32), r, -k], axis=1); radius[tang] = r
    origin[tang] = 0; d[tang] = (0, 0, -1)
    radius[110000:110100] = 0; d[110100:110200] = 0; d[110200:110300, 1] = np.nan
    ok_r, t0_r, t1_r = ref.sphere_hit(centre, radius, origin, d)
    ok_o = np.zeros(N, np.int32); t0_o = np.zeros(N, np.float32); t1_o = np.zeros(N, np.float32)
    oracle.lib.oracle_kat_sphere_hit(N, _p(centre), _p(radius), _p(origin), _p(d), _p(ok_o), _p(t0_o), _p(t1_o))
    assert np.array_equal(ok_r, ok_o)
    assert 0.25 < ok_r.mean() < 0.9 and np.all(ok_r[tang] == 1) and np.all(t0_r[tang] == t1_r[tang])
    hit = ok_r == 1
    assert_same_floats(t0_o, t0_r, "t0", hit); assert_same_floats(t1_o, t1_r, "t1", hit)
    # normalAt is p - centre
    n = np.zeros((1000, 3), np.float32)
    pts = origin[:1000]
    ref.lib.ref_sphere_normal_at(1000, _p(centre[:1000].copy()), _p(pts.copy()), _p(n))
    assert_same_floats(n, pts - centre[:1000], "normalAt")


def test_plane_hit(ref, oracle):
    rng = np.random.default_rng(13)
    p = _vectors(rng, N, 30); normal = _vectors(rng, N, 1); origin = _vectors(rng, N, 30); d = _vectors(rng, N, 1)
    normal[:1000] = (0, 1, 0); d[:1000, 1] = 0                    # parallel: division by zero
    d[1000:1100] = np.nan; normal[1100:1200] = 0
    ok_r, t_r, n_r = ref.plane_hit(p, normal, origin, d)
    ok_o = np.zeros(N, np.int32); t_o = np.zeros(N, np.float32)
    oracle.lib.oracle_kat_plane_hit(N, _p(p), _p(normal), _p(origin), _p(d), _p(ok_o), _p(t_o))
    assert np.array_equal(ok_r, ok_o) and 0.3 < ok_r.mean() < 0.7
    assert_same_floats(t_o, t_r, "t0")                              # Plane::hit writes t0 whether or not it reports a hit
    assert_same_floats(n_r, -normal, "normalAt")


def test_triangle_object_hit(ref, oracle):
    rng = np.random.default_rng(14)
    a = _vectors(rng, N, 10); b = a + _vectors(rng, N, 3); c = a + _vectors(rng, N, 3)
    w = rng.random((N, 2)).astype(np.float32)
    target = a + (b - a) * w[:, :1] * 0.7 + (c - a) * w[:, 1:] * 0.7
    origin = _vectors(rng, N, 20); d = f32(target - origin)
    miss = rng.random(N) < 0.3
    d[miss] = _vectors(rng, int(miss.sum()), 1)
    ok_r, t_r = ref.triangle_hit(a, b, c, origin, d)
    ok_o = np.zeros(N, np.int32); t_o = np.zeros(N, np.float32)
    oracle.lib.oracle_kat_triangle_object_hit(N, _p(a), _p(b), _p(c), _p(origin), _p(d), _p(ok_o), _p(t_o))
    assert np.array_equal(ok_r, ok_o) and 0.2 < ok_r.mean() < 0.9
    assert_same_floats(t_o, t_r, "t0", ok_r == 1)


def test_vec3_helpers(ref, oracle):
    rng = np.random.default_rng(15)
    a = _vectors(rng, N, 5); b = _vectors(rng, N, 5)
    a[:100] = 0; a[100:200] = 1e-30; a[200:300] = 3e19; b[300:400] = np.inf
    r = ref.vec3_ops(a, b)
    o = {"length": np.zeros(N, np.float32), "dot": np.zeros(N, np.float32), "cross": np.zeros((N, 3), np.float32),
         "normalise": np.zeros((N, 3), np.float32), "reflect": np.zeros((N, 3), np.float32), "radians": np.zeros(N, np.float32)}
    oracle.lib.oracle_kat_vec3_ops(N, _p(a), _p(b), _p(o["length"]), _p(o["dot"]), _p(o["cross"]), _p(o["normalise"]), _p(o["reflect"]),
                                   _p(o["radians"]))
    for key in o:
        assert_same_floats(o[key], r[key], key)


def _random_cameras(rng, n):
    pos = _vectors(rng, n, 50)
    fov = f32(rng.uniform(1, 179, n)); yaw = f32(rng.uniform(-720, 720, n)); pitch = f32(rng.uniform(-89, 89, n))
    fov[:4] = (90, 60, 179.9, 0.5); yaw[:6] = (-90, 0, 90, 180, -311.2, -93.5); pitch[:6] = (0, 89, -89, 13.8, 90, -90)
    return pos, fov, yaw, pitch


def test_camera_basis_oracle_and_product(ref, oracle):
    """Camera::Camera + updateDirection against the oracle's restatement AND the product's rt_camera_look (host code of the
    C ABI, no GPU involved)."""
    rng = np.random.default_rng(16)
    n = 20000
    pos, fov, yaw, pitch = _random_cameras(rng, n)
    for i in range(n):
        want = ref_lib.camera_fields(ref.camera(pos[i], float(fov[i]), float(yaw[i]), float(pitch[i])))
        for name, cam in (("oracle", oracle.camera_basis(pos[i], float(fov[i]), float(yaw[i]), float(pitch[i]))),
                          ("rt_camera_look", binding.camera_look(pos[i], float(fov[i]), float(yaw[i]), float(pitch[i])))):
            got = {"up": f32(cam.up.tuple()), "right": f32(cam.right.tuple()), "direction": f32(cam.direction.tuple()),
                   "position": f32(cam.position.tuple()), "fovFactor": f32(cam.fovFactor)}
            for key in got:
                assert np.array_equal(bits(got[key]), bits(want[key])), (name, key, i, got[key], want[key], fov[i], yaw[i], pitch[i])


@pytest.mark.parametrize("size", [(640, 480), (400, 150), (97, 211), (1, 1)])
def test_camera_rays(ref, oracle, size):
    """Per-pixel rasterToCameraSpace + normalise, pixel centres and the four anti-aliasing offsets, square or not (the
    reference ignores the aspect ratio, Camera.cpp:32)."""
    rng = np.random.default_rng(17)
    w, h = size
    pos, fov, yaw, pitch = _random_cameras(rng, 8)
    for i in range(8):
        cam84 = ref.camera(pos[i], float(fov[i]), float(yaw[i]), float(pitch[i]))
        cam = oracle.camera_basis(pos[i], float(fov[i]), float(yaw[i]), float(pitch[i]))
        for off in ((0.5, 0.5), (0.25, 0.25), (0.25, 0.75), (0.75, 0.25), (0.75, 0.75)):
            raw_r, dir_r = ref.camera_rays(cam84, w, h, offset=off)
            raw_o = np.zeros_like(raw_r); dir_o = np.zeros_like(dir_r)
            oracle.lib.oracle_kat_camera_rays(C.byref(cam), w, h, 0, 0, w, h, C.c_double(off[0]), C.c_double(off[1]), _p(raw_o), _p(dir_o))
            assert_same_floats(raw_o, raw_r, "rasterToCameraSpace"); assert_same_floats(dir_o, dir_r, "ray direction")


def test_camera_moves_of_the_host_mirror(ref, tmp_path):
    """Camera::move / mouseMovement: the C++ host mirror's camera, driven through its key handler, against the reference's."""
    import host_lib
    L = host_lib.lib()
    host_lib.write_models(str(tmp_path), ["Table", "Chalice", "Book2_Open", "Chest_gold"], 0.05)
    app = L.rth_create(32, 32, str(tmp_path).encode(), 0)
    try:
        assert L.rth_key_preset(app, b"B")              # scene 3: camera at (1.75159, -55.2005, 144.58), yaw -93.5, pitch 6.6
        cam = ref.camera((1.75159, -55.2005, 144.58), 90.0, -93.5, 6.6)
        keys = b"wwasd__|wdda s"
        move_of = {ord("w"): 0, ord("s"): 1, ord("a"): 2, ord("d"): 3, ord("_"): 4, ord("|"): 5}
        for k in keys:
            if k in move_of:
                L.rth_key_held(app, bytes([k]))
                cam = ref.camera_drive(cam, moves=[move_of[k]])
            got = T.Camera(); L.rth_camera(app, C.byref(got))
            want = ref_lib.camera_fields(cam)
            assert np.array_equal(bits(f32(got.position.tuple())), bits(want["position"])), (chr(k), got.position.tuple(), want["position"])
            assert np.array_equal(bits(f32(got.direction.tuple())), bits(want["direction"]))
    finally:
        L.rth_destroy(app)
    # mouse look: yaw / pitch accumulate in double and narrow, pitch clamps at +-89
    cam = ref.camera((0, 0, 0), 90.0, -90.0, 0.0)
    cam = ref.camera_drive(cam, mouse=[(13.0, -7.5), (1e3, 2e3), (-0.25, -1e4)])
    fields = ref_lib.camera_fields(cam)
    assert fields["pitch"] == f32(-89.0)
    yaw = f32(-90.0)
    for x in (13.0, 1e3, -0.25):
        yaw = f32(np.float64(yaw) + x * np.float64(f32(0.1)))
    assert fields["yaw"] == yaw
    want = binding.camera_look((0, 0, 0), 90.0, float(yaw), -89.0)
    assert np.array_equal(bits(f32(want.direction.tuple())), bits(fields["direction"])) and np.array_equal(bits(f32(want.up.tuple())), bits(fields["up"]))


def _triangle_cases(rng, n):
    v0 = _vectors(rng, n, 10); v1 = v0 + _vectors(rng, n, 3); v2 = v0 + _vectors(rng, n, 3)
    w = rng.random((n, 2)).astype(np.float32)
    target = v0 + (v1 - v0) * w[:, :1] * 0.8 + (v2 - v0) * w[:, 1:] * 0.8
    origin = _vectors(rng, n, 20); d = f32(target - origin)
    behind = rng.random(n) < 0.15; d[behind] = -d[behind]              # hits with t < 0: the branch without a box keeps them
    miss = rng.random(n) < 0.2; d[miss] = _vectors(rng, int(miss.sum()), 1)
    return np.concatenate([v0, v1, v2, origin, d], axis=1).astype(np.float32)


def _edge_triangle_cases():
    """det within a few units in the last place of the cull threshold -0.01 (a double literal), u and v on the edges."""
    cases = []
    s0 = f32(0.01)
    s = s0
    for _ in range(8):
        s = np.nextafter(s, f32(0))
    for _ in range(17):                                              # det = -s exactly: v0 = 0, v1 = (0,1,0), v2 = (s,0,0), d = -z
        cases.append([0, 0, -5, 0, 1, -5, float(s), 0, -5, float(s) / 4, 0.25, 0, 0, 0, -1])
        cases.append([0, 0, -5, float(s), 0, -5, 0, 1, -5, float(s) / 4, 0.25, 0, 0, 0, -1])      # det = +s
        s = np.nextafter(s, f32(1))
    for u, v in ((0, 0), (1, 0), (0, 1), (0.5, 0.5), (0.25, 0.75), (0, 0.5), (0.5, 0), (1.0000001, 0), (-1e-9, 0.5), (0.5, 0.50000006)):
        for flip in (False, True):
            p0, p1, p2 = np.float32([0, 0, -5]), np.float32([2, 0, -5]), np.float32([0, 2, -5])
            if flip:
                p1, p2 = p2, p1
            o = p0 + (p1 - p0) * f32(u) + (p2 - p0) * f32(v); o[2] = 0
            cases.append([*p0, *p1, *p2, *o, 0, 0, -1])
    cases.append([0, 0, -5, 1, 0, -5, 2, 0, -5, 0.5, 0, 0, 0, 0, -1])                              # degenerate: det = 0
    cases.append([0, 0, -5, 1, 0, -5, 0, 1, -5, 0.25, 0.25, 0, 0, 0, 0])                            # zero direction
    cases.append([0, 0, -5, 1, 0, -5, 0, 1, -5, 0.25, 0.25, 0, np.nan, 0, -1])
    return np.float32(cases)


@pytest.mark.parametrize("cull", [0, 1])
def test_triangle_intersect_through_model_hit(ref, oracle, cull):
    """Model::triangleIntersect (private) through Model::hit's branch without a box: one single-triangle model per case."""
    rng = np.random.default_rng(18 + cull)
    cases = np.concatenate([_edge_triangle_cases(), _triangle_cases(rng, N)])
    n = len(cases)
    hit_r, t_r, attrs_r = ref.triangle_cases(cases, cull)
    hit_o = np.zeros(n, np.int32); t_o = np.zeros(n, np.float32); attrs_o = np.zeros((n, 9), np.float32)
    oracle.lib.oracle_kat_triangle_cases(n, _p(cases), cull, _p(hit_o), _p(t_o), _p(attrs_o))
    assert np.array_equal(hit_r, hit_o)
    got = hit_r == 1
    assert (0.15 if cull else 0.3) < got.mean() < 0.9, got.mean()
    assert (t_r[got] < 0).mean() > 0.05                              # no `t > 0` test in this branch (Model.cpp:327)
    assert_same_floats(t_o, t_r, "t0", got)
    assert_same_floats(attrs_o, attrs_r, "(u, v, 1-u-v) and interpolated attributes")
    # the threshold cases: with culling only det < -0.01 (as a double) survives
    edge = len(_edge_triangle_cases())
    assert np.array_equal(hit_r[:edge], hit_o[:edge])
    if cull:
        s = np.abs(cases[0:34:2, 6].astype(np.float64))               # the negative-determinant cases, det = -s
        assert np.array_equal(hit_r[0:34:2] == 1, -s < -0.01) and 0 < (hit_r[0:34:2] == 1).sum() < 17
        assert not hit_r[1:34:2].any()                               # positive determinants never pass the cull


@pytest.mark.parametrize("name,cull", [("cube", 1), ("Chalice", 1), ("Chalice", 0), ("Pizza", 1)])
def test_model_hit_without_a_box_on_generated_meshes(ref, oracle, name, cull):
    """The whole loop of Model.cpp:318-356 on a baked mesh: closest-hit rule, dropped last triangle, interpolated normal /
    Ka / Kd — camera rays of a small frame plus random rays."""
    rng = np.random.default_rng(20)
    spec = scenes.SceneSpec(light=(0.0, -50.0, 0.0))
    spec.models = [scenes.ModelSpec(name, (0.0, 6.0, 10.0), 10, ((1, 1, 1), (1, 1, 1), 0.1, 0.9, 0.5, 100.0))]
    built = scenes.build_scene(spec, meshes={name: meshgen.generate(name, 0.15)})
    model = built.desc.models[0]
    nv, ni = model.vertCount, model.indicesCount
    verts = np.ctypeslib.as_array(C.cast(model.vertices, C.POINTER(C.c_float)), shape=(nv, 19)).copy()
    verts12 = np.concatenate([verts[:, 0:3], verts[:, 6:9], verts[:, 10:13], verts[:, 13:16]], axis=1)
    indices = np.ctypeslib.as_array(model.indices, shape=(ni,)).copy()
    cam = ref.camera((-11.2339, -4.81805, -13.431), 90.0, -311.2, 13.8)
    _, dirs = ref.camera_rays(cam, 96, 54)
    rays = np.concatenate([np.tile(f32([-11.2339, -4.81805, -13.431]), (96 * 54, 1)), dirs.reshape(-1, 3)], axis=1)
    centre = verts[:, 0:3].mean(axis=0)
    o = f32(centre + rng.normal(size=(3000, 3)) * 25)
    d = f32(centre + rng.normal(size=(3000, 3)) * 3 - o)
    rays = np.concatenate([rays, np.concatenate([o, d], axis=1)]).astype(np.float32)
    hit_r, t_r, attrs_r = ref.model_hit_nobox(verts12, indices, rays, cull)
    n = len(rays)
    hit_o = np.zeros(n, np.int32); t_o = np.zeros(n, np.float32); attrs_o = np.zeros((n, 9), np.float32)
    oracle.lib.oracle_kat_model_hit_nobox(nv, _p(verts12), ni, _p(indices), n, _p(rays), cull, _p(hit_o), _p(t_o), _p(attrs_o))
    assert np.array_equal(hit_r, hit_o) and hit_r.mean() > 0.05, hit_r.mean()
    assert_same_floats(t_o, t_r, "t0", hit_r == 1)
    assert_same_floats(attrs_o, attrs_r, "interpolated normal / Ka / Kd")


def test_last_triangle_is_dropped_by_the_reference_itself(ref):
    """Model.cpp:324 `i < indicesCount - 3`: two triangles, only the first can be hit, whichever it is."""
    a = [[-10, -10, -5], [0, 10, -5], [10, -10, -5]]
    b = [[-10, -10, -8], [0, 10, -8], [10, -10, -8]]
    ray = [[0, 0, 0, 0, 0, -1]]
    for tris, want in (([a], None), ([a, b], 5.0), ([b, a], 8.0), ([b, a, a], 5.0)):
        pts = np.float32(tris).reshape(-1, 3)
        verts12 = np.zeros((len(pts), 12), np.float32); verts12[:, 0:3] = pts
        hit, t, _ = ref.model_hit_nobox(verts12, np.arange(len(pts), dtype=np.int32), ray, 0)
        assert (hit[0] == 1) == (want is not None) and (want is None or t[0] == f32(want))
