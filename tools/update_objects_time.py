#!/usr/bin/env python3
"""Times rt_update_objects (new sphere / plane / box tables over the meshes already on the GPU) against a full rt_upload_scene
of the same scene.  Usage (GPU box): python tools/update_objects_time.py <config> [detail]"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

cfg_no = int(sys.argv[1]) if len(sys.argv) > 1 else 3
detail = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
spec, cfg, w, h = scenes.config_scene(cfg_no, detail=detail)
built = scenes.build_scene(spec)
r = binding.Renderer(0)
t0 = time.perf_counter(); r.upload(built.desc); t_upload = time.perf_counter() - t0
r.set_state(built.camera, cfg)
r.render(w, h)
spec2 = copy.copy(spec)
spec2.spheres = list(spec.spheres) + [((5.0, 0.0, -30.0), 4.0, ((0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 0.2, 0.8, 0.5, 50.0), False, T.RT_REFLECT)]
# the same model structs (their baked arrays are reused), one more sphere
d2 = T.SceneDesc()
C_ = __import__('ctypes')
C_.memmove(C_.byref(d2), C_.byref(built.desc), C_.sizeof(T.SceneDesc))
sph = (T.Sphere * len(spec2.spheres))()
for i, (pos, radius, mat, dbg, typ) in enumerate(spec2.spheres):
    sph[i] = T.Sphere(T.vec3(pos), radius, T.material(*mat), typ, int(bool(dbg)))
import ctypes as C
d2.spheres = C.cast(sph, C.POINTER(T.Sphere)); d2.sphereCount = len(spec2.spheres)
t0 = time.perf_counter(); r.update_objects(d2); t_update = time.perf_counter() - t0
a, _ = r.render(w, h)
t0 = time.perf_counter(); r.upload(d2); t_upload2 = time.perf_counter() - t0
b, _ = r.render(w, h)
import numpy as np
print("config %d detail %g: %d triangles; full upload %.1f ms (again with the sphere: %.1f ms), rt_update_objects %.2f ms; frames equal: %s" % (
    cfg_no, detail, r.scene_info().triangleCount, t_upload * 1e3, t_upload2 * 1e3, t_update * 1e3, bool(np.array_equal(a, b))))
