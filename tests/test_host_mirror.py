"""The C++ host mirror of the reference's scene API (Camera / Sphere / Plane / AABB / Model / objl::Loader /
RayTracer / presets), driven through a small C shim.  CPU tests compare what it hands to the C ABI with the Python
scene tables; the GPU test renders through it."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_package
import host_lib

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

REFERENCE_MODELS = ["Table", "Chalice", "Book2_Open", "Chest_gold"]


def _bytes(ptr, count, typ):
    return bytes((typ * count).from_address(C.addressof(ptr.contents))) if count else b""


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("models"))
    host_lib.write_models(d, REFERENCE_MODELS + ["cube"], 0.4)
    return d


@pytest.mark.parametrize("key", ["Z", "X", "C", "V", "B", "N", "M", ".", ","])
def test_scene_tables_presets_and_fixups_match(key, model_dir):
    L = host_lib.lib()
    w, h = 320, 180
    app = L.rth_create(w, h, model_dir.encode(), 0)     # on a machine without a GPU the context is missing; tables still build
    try:
        assert L.rth_key_preset(app, key.encode()) == 1
        desc = L.rth_describe(app).contents
        cam, cfg = T.Camera(), T.Config()
        L.rth_camera(app, cam); L.rth_config(app, cfg)

        index, pcfg = scenes.preset_config(key)
        spec = scenes.reference_scene(index, detail=0.4)
        scenes.apply_frame_fixups(spec, pcfg)
        built = scenes.build_scene(spec)
        assert bytes(cfg) == bytes(pcfg)
        assert bytes(cam) == bytes(built.camera)
        d = built.desc
        assert (desc.sphereCount, desc.planeCount, desc.AABBCount, desc.modelCount) == (d.sphereCount, d.planeCount, d.AABBCount, d.modelCount)
        assert _bytes(desc.spheres, d.sphereCount, T.Sphere) == _bytes(d.spheres, d.sphereCount, T.Sphere)
        assert _bytes(desc.planes, d.planeCount, T.Plane) == _bytes(d.planes, d.planeCount, T.Plane)
        assert _bytes(desc.AABBs, d.AABBCount, T.AABB) == _bytes(d.AABBs, d.AABBCount, T.AABB)
        assert _bytes(desc.lights, 1, T.Light) == _bytes(d.lights, 1, T.Light)
        for i in range(d.modelCount):
            a, b = desc.models[i], d.models[i]
            assert (a.vertCount, a.indicesCount, a.size, a.objectType, a.debug) == (b.vertCount, b.indicesCount, b.size, b.objectType, b.debug)
            assert bytes(a.mat) == bytes(b.mat) and bytes(a.bbMin) == bytes(b.bbMin) and bytes(a.bbMax) == bytes(b.bbMax)
            assert _bytes(a.vertices, a.vertCount, T.Vertex) == _bytes(b.vertices, b.vertCount, T.Vertex)
            assert _bytes(a.indices, a.indicesCount, C.c_int32) == _bytes(b.indices, b.indicesCount, C.c_int32)
            for k in range(8):
                ca, cb = a.boundingBoxes[k], b.boundingBoxes[k]
                assert bytes(ca.min) == bytes(cb.min) and bytes(ca.max) == bytes(cb.max) and ca.indicesCount == cb.indicesCount
                assert _bytes(ca.includedModelIndices, ca.indicesCount, C.c_int32) == _bytes(cb.includedModelIndices, cb.indicesCount, C.c_int32)
    finally:
        L.rth_destroy(app)


def test_presets_are_sticky_like_the_reference(model_dir):
    """Keys only set some fields; the rest keep what the previous preset left (Window.cpp:76-145)."""
    L = host_lib.lib()
    app = L.rth_create(64, 64, model_dir.encode(), 0)
    try:
        cfg = T.Config()
        L.rth_key_preset(app, b"X"); L.rth_key_preset(app, b"C"); L.rth_config(app, cfg)
        assert cfg.maxDepth == 1 and cfg.reflectPlanes == 1 and cfg.reflections == 0 and cfg.renderHardShadows == 1
        assert cfg.sphereReflectionStrength == np.float32(0.7)
        assert L.rth_key_preset(app, b"Q") == 0
    finally:
        L.rth_destroy(app)


def test_light_keys_move_light_and_marker_together(model_dir):
    L = host_lib.lib()
    app = L.rth_create(64, 64, model_dir.encode(), 0)
    try:
        L.rth_key_preset(app, b"C")
        for k in b"iiju":
            L.rth_key_held(app, bytes([k]))
        desc = L.rth_describe(app).contents
        assert desc.lights[0].position.tuple() == (-30.0, -31.0, -2.0)
        assert desc.spheres[0].position.tuple() == (-30.0, -31.0, -2.0)
    finally:
        L.rth_destroy(app)


def test_loader_quads_normals_negative_indices(tmp_path):
    L = host_lib.lib()
    # a unit quad (no normals -> face normal), a triangle with relative indices, and a second material group
    (tmp_path / "t.mtl").write_text("newmtl A\nKa 1 1 1\nKd 0.25 0.5 0.75\nKs 0.5 0.5 0.5\n\nnewmtl B\nKa 1 1 1\nKd 0.1 0.2 0.3\nKs 0 0 0\n")
    (tmp_path / "t.obj").write_text("mtllib t.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 2\nvn 0 0 1\n"
                                    "usemtl A\nf 1 2 3 4\nusemtl B\nf -1//1 -4//1 -3//1\nf 1//1 2//1 5//1\n")
    mat = T.material((1, 0, 0), (1, 0, 0), 0.1, 0.9, 0.0, 10.0)
    pos = (C.c_float * 3)(0, 0, 0)
    m = L.rth_model_load(pos, 1, str(tmp_path / "t.obj").encode(), mat, 0, 0)
    try:
        assert L.rth_model_loaded(m) == 1
        counts = (C.c_int * 4)(); L.rth_model_counts(m, counts)
        assert tuple(counts) == (10, 12, 2, 2)          # 4 + 3 + 3 corners; 2 + 1 + 1 triangles; 2 materials; 2 meshes
        view = T.Model(); L.rth_model_view(m, view)
        idx = [view.indices[i] for i in range(12)]
        assert idx[:9] == [0, 1, 3, 1, 2, 3, 4, 5, 6]   # quad -> (0,1,3), (1,2,3); the last triangle's indices are never copied
        assert idx[9:] == [0, 0, 0]
        v = [view.vertices[i] for i in range(10)]
        # face normal of the quad: cross(v0 - v1, v2 - v1) = cross((-1,0,0), (0,1,0)) = (0,0,-1); Y is mirrored by the bake
        assert v[0].normal.tuple() == (0.0, -0.0, -1.0)
        assert v[0].diffuse.tuple() == tuple(np.float32([0.25, 0.5, 0.75])) and v[4].diffuse.tuple() == tuple(np.float32([0.1, 0.2, 0.3]))
        assert v[4].position.tuple() == (0.0, -0.0, 2.0) and v[5].position.tuple() == (1.0, -0.0, 0.0) and v[6].position.tuple() == (1.0, -1.0, 0.0)
    finally:
        L.rth_model_free(m)
    out = (C.c_uint * 32)()
    tri = (C.c_float * 9)(0, 0, 0, 1, 0, 0, 0, 1, 0)
    assert L.rth_triangulate(tri, 3, out, 32) == 3 and list(out[:3]) == [0, 1, 2]
    pent = (C.c_float * 15)(0, 0, 0, 2, 0, 0, 3, 1, 0, 1, 3, 0, -1, 1, 0)
    n = L.rth_triangulate(pent, 5, out, 32)
    assert n == 9 and sorted(set(out[:n])) == [0, 1, 2, 3, 4]
    missing = L.rth_model_load(pos, 1, b"/nonexistent/x.obj", mat, 0, 0)
    assert L.rth_model_loaded(missing) == 0
    L.rth_model_free(missing)


def test_host_octree_holds_every_reachable_triangle(model_dir):
    L = host_lib.lib()
    mat = T.material((1, 0, 0), (1, 0, 0), 0.1, 0.9, 0.0, 10.0)
    pos = (C.c_float * 3)(0, 6, 10)
    m = L.rth_model_load(pos, 10, os.path.join(model_dir, "Chalice.obj").encode(), mat, 0, 1)
    try:
        counts = (C.c_int * 4)(); L.rth_model_counts(m, counts)
        stats = (C.c_int * 3)()
        L.rth_model_octree_stats(m, 8, 4, stats)
        assert stats[2] == counts[1] // 3 - 1 and stats[0] > stats[1] > 8
        L.rth_model_octree_stats(m, 1, 1000000, stats)      # everything fits one leaf: a root node with that leaf as its only child
        assert tuple(stats) == (9, 1, counts[1] // 3 - 1)
    finally:
        L.rth_model_free(m)


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["X", ".", "B"])
def test_host_mirror_renders_like_the_python_path(key, model_dir, renderer):
    L = host_lib.lib()
    w, h = 320, 180
    app = L.rth_create(w, h, model_dir.encode(), 0)
    try:
        L.rth_key_preset(app, key.encode())
        L.rth_display(app)
        L.rth_display(app)                               # second frame: scene already resident, only frame state moves
        frame = np.ctypeslib.as_array(L.rth_framebuffer(app), shape=(h, w, 3)).copy()
        rays = L.rth_last_rays(app)
    finally:
        L.rth_destroy(app)
    index, cfg = scenes.preset_config(key)
    spec = scenes.reference_scene(index, detail=0.4)
    scenes.apply_frame_fixups(spec, cfg)
    built = scenes.build_scene(spec)
    renderer.upload(built.desc)
    renderer.set_state(built.camera, cfg)
    rgb, st = renderer.render(w, h)
    assert np.array_equal(frame, rgb) and rays == st.rays


@pytest.mark.gpu
def test_reference_signature_entry_point(model_dir):
    """kernels.h:11's own argument list - rayTrace(width, height, framebuffer, scene, config, randStates) - over the default
    context: same bytes as RayTracer::launchKernel, across a scene switch (re-upload detected from the scene itself), a moved
    light (no re-upload) and an added sphere."""
    L = host_lib.lib()
    w, h = 320, 180
    app = L.rth_create(w, h, model_dir.encode(), 0)
    out = np.zeros((h, w, 3), np.uint8)
    try:
        for step in ("X", ".", "move", "add", "B"):
            if step == "move":
                L.rth_key_held(app, b"j")                  # moves the light and its marker sphere
            elif step == "add":
                mat = T.material((0, 1, 0), (0, 1, 0), 0.2, 0.8, 0.5, 50.0)
                L.rth_add_sphere(app, (C.c_float * 3)(5.0, 0.0, 5.0), 4.0, mat, 0)
            else:
                L.rth_key_preset(app, step.encode())
            L.rth_display(app)
            frame = np.ctypeslib.as_array(L.rth_framebuffer(app), shape=(h, w, 3)).copy()
            out[:] = 0
            assert L.rth_raytrace_reference(app, out.ctypes.data) == 0
            assert out.any() and np.array_equal(out, frame), step
    finally:
        L.rth_destroy(app)
        L.rth_raytrace_shutdown()


@pytest.mark.gpu
def test_headless_driver_sharded_path_writes_the_same_frame(model_dir, tmp_path):
    """rt_headless --sharded: the C++ host renders through rt_render_frame_sharded (this rank's tiles, RCCL gather to rank 0,
    reassembly) - with one GPU a one-rank communicator, which still runs every step of the exchange.  Same PPM as the plain run.
    (--gpus N forks one process per GPU and needs N GPUs; it shares all of this code but the fork.)"""
    import subprocess
    exe = os.path.join(os.path.dirname(host_lib.HOST_SO), "rt_headless")
    outs = []
    for extra in ([], ["--sharded"]):
        out = tmp_path / ("frame%d.ppm" % len(outs))
        cmd = [exe, "--size", "333x201", "--preset", ".", "--frames", "3", "--models", model_dir, "--out", str(out)] + extra
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr[-2000:]
        assert "Average render time" in res.stdout
        outs.append(out.read_bytes())
    assert len(outs[0]) > 333 * 201 * 3 and outs[0] == outs[1]


@pytest.mark.gpu
def test_headless_frame_loop_with_keys_writes_a_sequence(model_dir, tmp_path):
    """rt_headless --keys / --sequence: the reference's interactive loop (keys move camera and light between launches,
    Window.cpp:339-380) without the window; every frame lands on disk, consecutive frames differ, and the last frame of the
    sequence is the frame --out writes."""
    import subprocess
    exe = os.path.join(os.path.dirname(host_lib.HOST_SO), "rt_headless")
    seq = tmp_path / "seq"; seq.mkdir()
    out = tmp_path / "last.ppm"
    cmd = [exe, "--size", "200x120", "--preset", ".", "--frames", "6", "--models", model_dir, "--keys", "wd<jj ", "--sequence", str(seq), "--out", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    frames = [(seq / ("frame_%04d.ppm" % i)).read_bytes() for i in range(6)]
    assert all(len(f) > 200 * 120 * 3 for f in frames)
    assert frames[0] != frames[1] and frames[2] != frames[3] and frames[3] != frames[4]
    assert frames[4] == frames[5]                       # the blank key holds nothing: the same frame again
    assert frames[5] == out.read_bytes()


def test_view_without_a_display_reports_it_and_touches_no_gpu(tmp_path):
    """rt_view is the reference's window (Xlib + GLX + HIP's OpenGL interop).  No test machine has an X display, so what can
    be checked is the start-up contract: it opens the window before anything else, says why it cannot and exits with 3."""
    import subprocess
    exe = os.path.join(os.path.dirname(host_lib.HOST_SO), "rt_view")
    if not os.path.exists(exe):
        pytest.skip("rt_view is only built where the X11 and OpenGL headers are installed")
    env = {k: v for k, v in os.environ.items() if k not in ("DISPLAY", "WAYLAND_DISPLAY")}
    res = subprocess.run([exe, "--size", "64x48", "--frames", "1"], capture_output=True, text=True, timeout=60, env=env)
    assert res.returncode == 3, (res.returncode, res.stderr[-500:])
    assert "cannot open the X display" in res.stderr
    res = subprocess.run([exe, "--size", "64"], capture_output=True, text=True, timeout=60, env=env)
    assert res.returncode == 2
