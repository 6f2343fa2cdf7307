"""ctypes wrapper of oracle/liboracle.so (the CPU restatement of the reference; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import ROOT, load_package

load_package()
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle():
    # make knows what the library depends on (oracle.c and the C ABI's struct definitions); without a compiler at hand the
    # prebuilt library is used as it is
    try:
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(ORACLE_SO):
            raise
    return ORACLE_SO


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        vp, i32 = C.c_void_p, C.c_int
        L.oracle_render.argtypes = [C.POINTER(T.SceneDesc), C.POINTER(T.Camera), C.POINTER(T.Config), i32, i32, i32, i32, i32, i32,
                                    vp, vp, vp, vp, i32]
        L.oracle_render.restype = None
        L.oracle_solve_quadratic.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_wang_hash.argtypes = [C.c_uint32]
        L.oracle_wang_hash.restype = C.c_float
        L.oracle_sphere_hit.argtypes = [C.POINTER(T.Sphere), vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_plane_hit.argtypes = [C.POINTER(T.Plane), vp, vp, C.POINTER(C.c_float)]
        L.oracle_aabb_hit.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_aabb_normal_at.argtypes = [vp, vp, vp, vp]
        L.oracle_aabb_normal_at.restype = None
        L.oracle_triangle_intersect.argtypes = [vp, vp, vp, vp, vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_triangle_object_hit.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_float)]
        L.oracle_camera_ray.argtypes = [C.POINTER(T.Camera), C.c_float, C.c_float, i32, i32, vp]
        L.oracle_camera_ray.restype = None
        L.oracle_lighting.argtypes = [C.POINTER(T.Material), vp, vp, vp, vp, vp, C.POINTER(T.Config), i32, i32, vp]
        L.oracle_lighting.restype = None
        L.oracle_pack_channel.argtypes = [C.c_float]
        L.oracle_pack_channel.restype = C.c_uint8
        L.oracle_camera_basis.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.POINTER(T.Camera)]
        L.oracle_camera_basis.restype = None
        L.oracle_bake_model.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
        L.oracle_bake_model.restype = None

    def render(self, desc, camera, config, width, height, window=None, threads=8, want_ids=True):
        """Renders the window (x0, y0, x1, y1) (default: the full frame).  Pixels outside stay 0.
        Returns (rgb [h,w,3] u8, ids [h,w,2] i32, t [h,w] f32, rays, tri_tests)."""
        x0, y0, x1, y1 = window if window else (0, 0, width, height)
        rgb = np.zeros((height, width, 3), np.uint8)
        ids = np.full((height, width, 2), -1, np.int32) if want_ids else None
        t = np.full((height, width), 999.0, np.float32) if want_ids else None
        counters = np.zeros(2, np.uint64)
        self.lib.oracle_render(C.byref(desc), C.byref(camera), C.byref(config), width, height, x0, y0, x1, y1,
                               rgb.ctypes.data, ids.ctypes.data if want_ids else None, t.ctypes.data if want_ids else None,
                               counters.ctypes.data, threads)
        return rgb, ids, t, int(counters[0]), int(counters[1])

    def camera_basis(self, position, fov, yaw, pitch):
        cam = T.Camera()
        self.lib.oracle_camera_basis(_f3(position), fov, yaw, pitch, C.byref(cam))
        return cam
