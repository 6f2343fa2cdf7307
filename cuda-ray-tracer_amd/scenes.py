"""The reference's scene tables, key presets and per-frame fix-ups as data, plus the benchmark scenes.

Sources (reference checkout): scene tables RayTracer.cpp:71-337, SceneConfig defaults RayTracer.h:36-70,
key presets Window.cpp:76-145, per-frame fix-ups Window.cpp:382-384.

A `SceneSpec` is plain Python data; `build_scene` turns it into the ctypes structs of the C ABI (keeping the
backing arrays alive) and bakes models through the library's own `rt_bake_model`.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import binding, meshgen
from . import ctypes_types as T

f32 = np.float32


# ----------------------------------------------------------------------------- specs

@dataclass
class ModelSpec:
    name: str
    position: tuple
    size: int
    mat: tuple            # (ka, kd, ambient, diffuse, specular, shininess)
    debug: bool = False
    object_type: int = T.RT_DIFFUSE
    detail: float = 1.0


@dataclass
class SceneSpec:
    spheres: list = field(default_factory=list)   # (pos, radius, mat, debug, type)
    planes: list = field(default_factory=list)    # (pos, n, mat, type)
    boxes: list = field(default_factory=list)     # (centre, size, mat, debug, type)
    models: list = field(default_factory=list)
    light: tuple = (0.0, -40.0, -50.0)
    light_colour: tuple = (1.0, 1.0, 1.0)
    cam_pos: tuple = (0.0, 0.0, 0.0)
    cam_yaw: float = -90.0
    cam_pitch: float = 0.0
    fov: float = 90.0


WHITE = ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 1.0, 0.0, 0.0, 200.0)       # the light-marker sphere material
FLOOR = ((0.1, 0.1, 0.1), (0.1, 0.1, 0.1), 0.5, 0.5, 0.0, 200.0)


def _marker(pos):
    return (pos, 2.0, WHITE, True, T.RT_DIFFUSE)


def _floor():
    return ((0.0, 20.0, 0.0), (0.0, 1.0, 0.0), FLOOR, T.RT_REFLECT)


_MODEL_MAT_DIFFUSE = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 0.0, 10.0)
_MODEL_MAT_CHALICE = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 1.0, 1000.0)


def _table_scene_models(detail):
    return [ModelSpec("Table", (0.0, 19.0, 0.0), 15, _MODEL_MAT_DIFFUSE, False, T.RT_DIFFUSE, detail),
            ModelSpec("Chalice", (0.0, 6.0, 10.0), 10, _MODEL_MAT_CHALICE, False, T.RT_REFLECT, detail),
            ModelSpec("Book2_Open", (0.0, 6.0, 0.0), 10, _MODEL_MAT_DIFFUSE, False, T.RT_DIFFUSE, detail)]


def reference_scene(index, detail=1.0):
    """Scene `index` (0-8) of RayTracer::initialiseScenes."""
    R, D = T.RT_REFLECT, T.RT_DIFFUSE
    s = SceneSpec(planes=[_floor()])
    if index == 0:      # RayTracer.cpp:73-95
        s.cam_pos = (0.0, 0.0, 50.0); s.light = (0.0, -40.0, -50.0)
        s.spheres = [_marker((0.0, 0.0, -50.0))]
    elif index == 1:    # RayTracer.cpp:98-124
        s.light = (-30.0, -70.0, 0.0)
        s.spheres = [_marker((-30.0, -70.0, 0.0)),
                     ((0.0, 0.0, -70.0), 15.0, ((1, 0, 0), (1, 0, 0), 0.2, 0.2, 0.0, 200.0), False, R),
                     ((50.0, 0.0, -70.0), 15.0, ((0, 1, 0), (0, 1, 0), 0.2, 0.2, 0.0, 200.0), False, R),
                     ((0.0, -100.0, -240.0), 100.0, ((0, 1, 1), (0, 1, 0), 0.2, 0.2, 0.0, 200.0), False, R)]
        s.boxes = [((-50.0, 0.0, -70.0), 15.0, ((1, 0, 1), (1, 0, 1), 0.2, 0.2, 0.0, 200.0), False, R)]
    elif index == 2:    # RayTracer.cpp:127-157
        s.cam_pos = (-50.0, 0.0, 0.0); s.light = (0.0, -40.0, -50.0)
        s.spheres = [_marker((0.0, 0.0, -50.0))]
        red = ((1, 0, 0), (1, 0, 0), 0.1, 0.9, 0.9, 200.0)
        for i in range(10):
            for j in range(10):
                s.spheres.append(((j * 15.0, 0.0, float(i * 15)), 2.0, red, False, D))
    elif index == 3:    # RayTracer.cpp:160-198
        s.cam_pos = (1.75159, -55.2005, 144.58); s.cam_yaw = -93.5; s.cam_pitch = 6.6; s.light = (-30.0, -70.0, 0.0)
        s.spheres = [_marker((-30.0, -70.0, 0.0))]
        blue = ((0, 0, 1), (0, 0, 1), 0.2, 0.2, 0.3, 10.0)
        yellow = ((1, 1, 0), (1, 1, 0), 0.2, 0.2, 0.3, 10.0)
        cyan = ((0, 1, 1), (0, 1, 1), 0.2, 0.2, 0.3, 10.0)
        by_slot = {0: ((100, 0, 0), 100.0, blue, R), 3: ((100, -100, 0), 100.0, blue, R), 8: ((100, 0, 100), 100.0, blue, R),
                   9: ((100, -100, 100), 100.0, blue, R), 1: ((-100, 0, 0), 100.0, yellow, R), 4: ((-100, -100, 0), 100.0, yellow, R),
                   10: ((-100, 0, 100), 100.0, yellow, R), 11: ((-100, -100, 100), 100.0, yellow, R),
                   2: ((0, 0, -100), 100.0, cyan, D), 5: ((0, -100, -100), 100.0, cyan, D),
                   6: ((0, 0, 10), 40.0, ((1, 0, 0), (1, 0, 0), 0.2, 0.8, 0.9, 200.0), D),
                   7: ((0, -30, 10), 20.0, ((0, 1, 0), (0, 1, 0), 0.2, 0.8, 0.9, 200.0), D)}
        s.boxes = [(by_slot[k][0], by_slot[k][1], by_slot[k][2], False, by_slot[k][3]) for k in range(12)]
    elif index == 4:    # RayTracer.cpp:201-226
        s.cam_pos = (-11.2339, -4.81805, -13.431); s.cam_yaw = -311.2; s.cam_pitch = 13.8; s.light = (-30.0, -70.0, 0.0)
        s.spheres = [_marker((-30.0, -70.0, 0.0))]
        s.models = _table_scene_models(detail)
    elif index == 5:    # RayTracer.cpp:229-254
        s.light = (-29.0, -32.0, 0.0)
        s.spheres = [_marker((-29.0, -32.0, 0.0)),
                     ((0.0, 0.0, -70.0), 15.0, ((1, 0, 1), (1, 0, 1), 0.2, 0.8, 0.8, 200.0), False, D),
                     ((20.0, 0.0, -120.0), 15.0, ((0, 1, 1), (0, 1, 1), 0.2, 0.8, 0.8, 200.0), False, D),
                     ((-20.0, 0.0, -120.0), 15.0, ((1, 1, 0), (1, 1, 0), 0.2, 0.8, 0.8, 200.0), False, D)]
    elif index == 6:    # RayTracer.cpp:257-280
        s.cam_pos = (48.0561, -33.9654, 32.7296); s.cam_yaw = -146.2; s.cam_pitch = 19.7; s.light = (21.0, -13.0, -18.0)
        s.spheres = [_marker((21.0, -13.0, -18.0))]
        s.models = [ModelSpec("Chest_gold", (0.0, 0.0, 0.0), 15, ((1, 0, 0), (1, 0, 0), 0.1, 0.9, 0.9, 200.0), False, D, detail)]
    elif index == 7:    # RayTracer.cpp:282-309
        s.cam_pos = (0.774152, -77.2231, 82.8138); s.cam_yaw = -89.5; s.cam_pitch = 28.7; s.light = (2.0, -73.0, -16.0)
        pink = ((1, 0, 1), (1, 0, 1), 0.2, 0.8, 0.8, 200.0)
        s.spheres = [_marker((2.0, -73.0, -16.0)), ((0, 0, 0), 15.0, pink, False, R), ((0, 0, -40.0), 15.0, pink, False, R),
                     ((0, 0, 40.0), 15.0, pink, False, R)]
        grey = ((0.2, 0.2, 0.2), (0.2, 0.2, 0.2), 0.5, 0.2, 0.3, 10.0)
        s.boxes = [((-100.0, 0, 0), 100.0, grey, False, R), ((100.0, 0, 0), 100.0, grey, False, R)]
    elif index == 8:    # RayTracer.cpp:311-337
        s.cam_pos = (38.9299, -39.5555, 76.1236); s.cam_yaw = -129.6; s.cam_pitch = 15.0; s.light = (2.0, -73.0, -16.0)
        s.spheres = [_marker((2.0, -73.0, -16.0))]
        s.boxes = [((-100.0, 0, 0), 100.0, ((0.2, 0.2, 0.2), (0.2, 0.2, 0.2), 0.5, 0.2, 0.3, 10.0), False, R)]
        s.models = _table_scene_models(detail)
    else:
        raise ValueError("scene index 0..8")
    return s


def default_config():
    """SceneConfig defaults, RayTracer.h:36-70."""
    c = T.Config()
    c.fps = 0
    c.shadowBias = 0.01; c.renderHardShadows = 0; c.renderSoftShadows = 0; c.softShadowRadius = 5; c.softShadowNum = 15
    c.shadowIntensity = 0.5; c.areaLightSpecularEffect = 0; c.reflections = 0; c.maxDepth = 2; c.reflectPlanes = 0
    c.sphereReflectionStrength = 0.5; c.planeReflectionStrength = 0.5; c.AABBReflectionStrength = 0.5; c.modelReflectionStrength = 0.5
    c.renderAABBs = 1; c.renderModels = 1; c.ambientLighting = 1; c.diffuseLighting = 1; c.specularLighting = 1
    c.antiAliasing = 0; c.boundingBox = 1; c.eightBoundingBoxes = 0; c.octree = 0; c.cullBackTriangles = 1
    c.backgroundCol = T.Vec3(0.1, 0.1, 0.1); c.backgroundBrightness = 7; c.floorBrightness = 3
    return c


# key -> (scene index, {field: value}); Window.cpp:76-145
PRESETS = {
    "Z": (0, {}),
    "X": (1, dict(maxDepth=1, renderHardShadows=0, renderSoftShadows=0, reflections=1, reflectPlanes=1,
                  planeReflectionStrength=0.5, sphereReflectionStrength=0.7)),
    "C": (5, dict(reflections=0, renderHardShadows=1)),
    "V": (2, dict(renderHardShadows=0, renderSoftShadows=0, reflections=0)),
    "B": (3, dict(reflections=1, planeReflectionStrength=0.22, renderHardShadows=0, renderSoftShadows=0)),
    "N": (7, dict(renderHardShadows=0, renderSoftShadows=0, reflections=1, reflectPlanes=1, maxDepth=1, planeReflectionStrength=0.5)),
    "M": (6, dict(renderHardShadows=0, renderSoftShadows=0, reflections=0)),
    ",": (8, dict(reflections=1, reflectPlanes=1, planeReflectionStrength=0.5)),
    ".": (4, dict(renderHardShadows=1, renderSoftShadows=0, reflections=1, modelReflectionStrength=0.8)),
}


def preset_config(key, base=None):
    """Defaults (or `base`) with exactly one key preset applied; returns (scene index, config)."""
    cfg = base if base is not None else default_config()
    index, fields = PRESETS[key]
    for k, v in fields.items():
        setattr(cfg, k, v)
    return index, cfg


def apply_frame_fixups(spec, cfg):
    """What Window::display rewrites before every launch (Window.cpp:382-384).  Mutates spec and cfg."""
    if spec.spheres:
        pos, _r, mat, dbg, typ = spec.spheres[0]
        spec.spheres[0] = (pos, float(cfg.softShadowRadius), mat, dbg, typ)
    if spec.planes:
        pos, n, mat, typ = spec.planes[0]
        v = float(f32(0.1) * f32(cfg.floorBrightness))
        spec.planes[0] = (pos, n, ((v, v, v),) + tuple(mat[1:]), typ)
    b = float(f32(0.1) * f32(cfg.backgroundBrightness))
    cfg.backgroundCol = T.Vec3(b, b, b)


# ----------------------------------------------------------------------------- spec -> C ABI structs

class BuiltScene:
    """ctypes structs for one scene plus everything they point to."""

    def __init__(self):
        self.desc = T.SceneDesc()
        self.camera = None
        self.keep = []          # backing arrays
        self.model_meshes = []  # LoadedMesh per model (tests use them)


def bake_model(mesh, position, size):
    """Runs the library's Model-constructor equivalent on loader-shaped arrays.  Returns a dict of numpy arrays."""
    L = binding.lib()
    n, ni = mesh.positions.shape[0], mesh.indices.shape[0]
    per_box = 3 * int(np.ceil((ni // 3) / 8.0))
    verts = (T.Vertex * max(n, 1))()
    indices = np.zeros(max(ni, 1), np.int32)
    bb = np.zeros(6, np.float32)
    chunk_boxes = np.zeros(48, np.float32)
    chunk_idx = np.zeros(max(8 * per_box, 1), np.int32)
    got = C.c_int32(0)
    pos3 = (C.c_float * 3)(*[float(v) for v in position])
    arrs = [np.ascontiguousarray(a, np.float32) for a in (mesh.positions, mesh.normals, mesh.texcoords, mesh.ka, mesh.kd, mesh.ks)]
    idx = np.ascontiguousarray(mesh.indices, np.uint32)
    rc = L.rt_bake_model(*[a.ctypes.data for a in arrs], n, idx.ctypes.data, ni, pos3, int(size),
                         C.addressof(verts), indices.ctypes.data, bb.ctypes.data, chunk_boxes.ctypes.data,
                         chunk_idx.ctypes.data, C.byref(got))
    if rc != 0:
        raise binding.RtError(f"rt_bake_model failed ({rc})")
    assert got.value == per_box
    return dict(vertices=verts, vert_count=n, indices=indices, indices_count=ni, bb=bb, chunk_boxes=chunk_boxes,
                chunk_idx=chunk_idx, per_box=per_box)


def _model_struct(baked, position, size, mat, debug, object_type, keep):
    m = T.Model()
    m.position = T.vec3(position); m.size = int(size); m.mat = T.material(*mat)
    m.objectType = int(object_type); m.debug = int(bool(debug))
    bb = baked["bb"]
    m.bbMin = T.Vec3(*bb[:3]); m.bbMax = T.Vec3(*bb[3:])
    per_box = baked["per_box"]
    for b in range(8):
        cb = baked["chunk_boxes"][6 * b:6 * b + 6]
        m.boundingBoxes[b].min = T.Vec3(*cb[:3]); m.boundingBoxes[b].max = T.Vec3(*cb[3:])
        m.boundingBoxes[b].indicesCount = per_box
        sub = baked["chunk_idx"][b * per_box:(b + 1) * per_box]
        m.boundingBoxes[b].includedModelIndices = sub.ctypes.data_as(C.POINTER(C.c_int32)) if per_box else None
    m.vertices = C.cast(baked["vertices"], C.POINTER(T.Vertex)); m.vertCount = baked["vert_count"]
    m.indices = baked["indices"].ctypes.data_as(C.POINTER(C.c_int32)); m.indicesCount = baked["indices_count"]
    keep.append(baked)
    return m


_MESH_CACHE = {}


def get_mesh(name, detail):
    key = (name, float(detail))
    if key not in _MESH_CACHE:
        _MESH_CACHE[key] = meshgen.generate(name, detail)
    return _MESH_CACHE[key]


def build_scene(spec, octree_max_depth=0, octree_leaf_triangles=0, meshes=None, build_on_device=False):
    """SceneSpec -> BuiltScene.  `meshes` may map model name -> LoadedMesh to override the generators;
    build_on_device asks rt_upload_scene to construct the octrees on the GPU."""
    out = BuiltScene()
    d = out.desc
    sph = (T.Sphere * max(len(spec.spheres), 1))()
    for i, (pos, radius, mat, dbg, typ) in enumerate(spec.spheres):
        sph[i] = T.Sphere(T.vec3(pos), radius, T.material(*mat), typ, int(bool(dbg)))
    pln = (T.Plane * max(len(spec.planes), 1))()
    for i, (pos, n, mat, typ) in enumerate(spec.planes):
        pln[i] = T.Plane(T.vec3(pos), T.vec3(n), T.material(*mat), typ, 0)      # Plane.cpp:11 forces debug off
    box = (T.AABB * max(len(spec.boxes), 1))()
    for i, (centre, size, mat, dbg, typ) in enumerate(spec.boxes):
        half = f32(size) / f32(2)                                               # AABB.cpp:8-10
        lo = [float(f32(c) - half) for c in centre]
        hi = [float(f32(c) + half) for c in centre]
        box[i] = T.AABB(T.vec3(lo), T.vec3(hi), T.material(*mat), typ, int(bool(dbg)))
    mdl = (T.Model * max(len(spec.models), 1))()
    for i, ms in enumerate(spec.models):
        mesh = meshes[ms.name] if meshes and ms.name in meshes else get_mesh(ms.name, ms.detail)
        baked = bake_model(mesh, ms.position, ms.size)
        mdl[i] = _model_struct(baked, ms.position, ms.size, ms.mat, ms.debug, ms.object_type, out.keep)
        out.model_meshes.append(mesh)
    light = (T.Light * 1)()
    light[0] = T.Light(T.vec3(spec.light), T.vec3(spec.light_colour))
    d.spheres = C.cast(sph, C.POINTER(T.Sphere)); d.sphereCount = len(spec.spheres)
    d.planes = C.cast(pln, C.POINTER(T.Plane)); d.planeCount = len(spec.planes)
    d.AABBs = C.cast(box, C.POINTER(T.AABB)); d.AABBCount = len(spec.boxes)
    d.models = C.cast(mdl, C.POINTER(T.Model)); d.modelCount = len(spec.models)
    d.lights = C.cast(light, C.POINTER(T.Light)); d.lightCount = 1
    d.octreeMaxDepth = octree_max_depth; d.octreeLeafTriangles = octree_leaf_triangles
    d.buildOnDevice = 1 if build_on_device else 0
    out.keep += [sph, pln, box, mdl, light]
    d._keepalive = out.keep      # the descriptor alone keeps its arrays alive
    out.camera = binding.camera_look(spec.cam_pos, spec.fov, spec.cam_yaw, spec.cam_pitch)
    return out


# ----------------------------------------------------------------------------- benchmark configurations

def config_scene(number, detail=1.0):
    """The benchmark configurations: returns (SceneSpec, Config, width, height)."""
    if number == 1:     # reference scene 1 + the X preset at 640x480
        index, cfg = preset_config("X")
        spec = reference_scene(index)
        w, h = 640, 480
    elif number == 2:   # cube + Chalice on the table scene's floor and light, hard shadows, no reflections
        cfg = default_config(); cfg.renderHardShadows = 1; cfg.reflections = 0
        spec = SceneSpec(planes=[_floor()], light=(-30.0, -70.0, 0.0), spheres=[_marker((-30.0, -70.0, 0.0))],
                         cam_pos=(-11.2339, -4.81805, -13.431), cam_yaw=-311.2, cam_pitch=13.8)
        spec.models = [ModelSpec("cube", (-6.0, 17.0, 6.0), 3, _MODEL_MAT_DIFFUSE, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Chalice", (0.0, 20.0, 10.0), 10, _MODEL_MAT_CHALICE, False, T.RT_REFLECT, detail)]
        w, h = 1280, 720
    elif number == 3:   # Crown + Vase + Crystal3, hard shadows, reflections (maxDepth 2), reflective floor
        cfg = default_config(); cfg.renderHardShadows = 1; cfg.reflections = 1; cfg.reflectPlanes = 1; cfg.maxDepth = 2
        cfg.modelReflectionStrength = 0.8
        spec = SceneSpec(planes=[_floor()], light=(-30.0, -70.0, 40.0), spheres=[_marker((-30.0, -70.0, 40.0))],
                         cam_pos=(0.0, -22.0, 92.0), cam_yaw=-90.0, cam_pitch=8.0)
        gold = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 1.0, 1000.0)
        matte = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 0.3, 50.0)
        spec.models = [ModelSpec("Crown", (-46.0, 20.0, -8.0), 30, gold, False, T.RT_REFLECT, detail),
                       ModelSpec("Vase", (0.0, 20.0, 0.0), 32, matte, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Crystal3", (46.0, 20.0, -4.0), 36, gold, False, T.RT_REFLECT, detail)]
        w, h = 1920, 1080
    elif number == 4:   # multi-model room
        cfg = default_config(); cfg.renderHardShadows = 1; cfg.reflections = 1; cfg.reflectPlanes = 1
        spec = SceneSpec(planes=[_floor()], light=(-30.0, -90.0, 60.0), spheres=[_marker((-30.0, -90.0, 60.0))],
                         cam_pos=(0.0, -40.0, 130.0), cam_yaw=-90.0, cam_pitch=10.0)
        d = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 0.2, 20.0)
        shiny = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 1.0, 500.0)
        spec.models = [ModelSpec("Table", (0.0, 20.0, 0.0), 30, d, False, T.RT_DIFFUSE, detail),
                       ModelSpec("BookCaseLargeBooks", (-80.0, 20.0, -30.0), 30, d, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Chest_gold", (70.0, 20.0, -10.0), 25, shiny, False, T.RT_REFLECT, detail),
                       ModelSpec("Pizza", (-8.0, -10.0, 0.0), 8, d, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Pug", (35.0, 20.0, 40.0), 18, d, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Chalice", (10.0, -10.0, 5.0), 8, shiny, False, T.RT_REFLECT, detail),
                       ModelSpec("Vase", (-40.0, 20.0, 35.0), 20, d, False, T.RT_DIFFUSE, detail),
                       ModelSpec("Crown", (12.0, -10.0, -8.0), 8, shiny, False, T.RT_REFLECT, detail),
                       ModelSpec("Book2_Open", (-10.0, -10.0, 10.0), 10, d, False, T.RT_DIFFUSE, detail)]
        w, h = 3840, 2160
    elif number == 5:   # field of baked copies (the reference has no instancing) of Chicken and WoodLog, ~10 M triangles at detail 8
        cfg = default_config(); cfg.renderHardShadows = 1; cfg.reflections = 1; cfg.reflectPlanes = 1
        spec = SceneSpec(planes=[_floor()], light=(-60.0, -160.0, 120.0), spheres=[_marker((-60.0, -160.0, 120.0))],
                         cam_pos=(0.0, -70.0, 260.0), cam_yaw=-90.0, cam_pitch=14.0)
        matte = ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.1, 0.9, 0.2, 20.0)
        n = 12
        for i in range(n):
            for j in range(n):
                name = "Chicken" if (i + j) % 2 == 0 else "WoodLog"
                x = (j - (n - 1) / 2.0) * 34.0
                z = 120.0 - i * 34.0
                spec.models.append(ModelSpec(name, (x, 20.0, z), 14, matte, False, T.RT_DIFFUSE, detail))
        w, h = 7680, 4320
    else:
        raise ValueError("configuration 1..5")
    apply_frame_fixups(spec, cfg)
    return spec, cfg, w, h


def camera_path_pose(spec, frame, frames=120):
    """The fixed camera path of the 120-frame benchmark: a yaw sweep of +-12 degrees with a slow dolly towards the scene.
    Returns (position, yaw, pitch) for `frame`."""
    import math
    u = frame / max(frames - 1, 1)
    yaw = spec.cam_yaw + 12.0 * math.sin(2.0 * math.pi * u)
    pos = (spec.cam_pos[0] + 20.0 * math.sin(2.0 * math.pi * u), spec.cam_pos[1], spec.cam_pos[2] - 40.0 * u)
    return pos, yaw, spec.cam_pitch
