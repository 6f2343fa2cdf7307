"""Access to oracle/_ref/: the reference's own sources compiled in this image (oracle/build_ref.sh), used to pin the CPU
oracle, the host loader and - through fixtures made from it - the HIP kernels.  Test infrastructure only.

`available()` is False where neither the prebuilt files nor the reference checkout exist; tests that need it then skip.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

from conftest import ROOT

ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_SO = os.path.join(REF_DIR, "libref.so")
REF_MODEL = os.path.join(REF_DIR, "ref_model_kat")
REFERENCE_CHECKOUT = os.environ.get("RT_REFERENCE_DIR", "/root/reference")

_built = None


def build():
    """Runs the recipe when the reference checkout is present (it is cheap: a few seconds); returns whether _ref exists."""
    global _built
    if _built is None:
        sources = [os.path.join(ORACLE_DIR, f) for f in ("build_ref.sh", "ref_driver.cpp", "ref_model_driver.cpp")]
        have = os.path.exists(REF_SO) and os.path.exists(REF_MODEL)
        stale = have and min(os.path.getmtime(REF_SO), os.path.getmtime(REF_MODEL)) < max(os.path.getmtime(s) for s in sources)
        if os.path.exists(os.path.join(REFERENCE_CHECKOUT, "kernels.cu")) and (not have or stale):
            try:
                subprocess.check_call(["bash", sources[0]], stdout=subprocess.DEVNULL)
            except (OSError, subprocess.CalledProcessError):
                pass
        _built = os.path.exists(REF_SO) and os.path.exists(REF_MODEL)
    return _built


def available():
    return build()


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


CAMERA_BYTES = 84      # sizeof(Camera) in the reference: up, right, direction, position, fov, fovFactor, aspectRatio, near, far, moveSpeed, sensitivity, pitch, yaw


def camera_fields(cam84):
    """Decodes the reference's Camera object: dict of up, right, direction, position (3 floats each), fovFactor, pitch, yaw."""
    f = np.frombuffer(bytes(cam84), np.float32)
    return {"up": f[0:3].copy(), "right": f[3:6].copy(), "direction": f[6:9].copy(), "position": f[9:12].copy(), "fov": f[12],
            "fovFactor": f[13], "pitch": f[19], "yaw": f[20]}


class Ref:
    def __init__(self):
        if not build():
            raise RuntimeError("oracle/_ref is not built and the reference checkout is absent")
        self.lib = L = C.CDLL(REF_SO)
        vp, i32, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
        L.ref_sizes.argtypes = [vp]
        L.ref_solve_quadratic.argtypes = [i32] + [vp] * 6
        L.ref_sphere_hit.argtypes = [i32] + [vp] * 7
        L.ref_sphere_normal_at.argtypes = [i32] + [vp] * 3
        L.ref_plane_hit.argtypes = [i32] + [vp] * 7
        L.ref_triangle_hit.argtypes = [i32] + [vp] * 7
        L.ref_vec3_ops.argtypes = [i32] + [vp] * 8
        L.ref_camera_make.argtypes = [vp, f, f, f, f, vp]
        L.ref_camera_drive.argtypes = [vp, vp, i32, vp, i32]
        L.ref_camera_rays.argtypes = [vp, i32, i32, i32, i32, i32, i32, d, d, vp, vp]
        L.ref_frame_spheres.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        L.ref_frame_planes.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        L.ref_loader_load.argtypes = [C.c_char_p]; L.ref_loader_load.restype = vp
        L.ref_loader_free.argtypes = [vp]; L.ref_loader_free.restype = None
        L.ref_loader_counts.argtypes = [vp, C.POINTER(i32)]; L.ref_loader_counts.restype = None
        L.ref_loader_arrays.argtypes = [vp, vp, vp]; L.ref_loader_arrays.restype = None
        L.ref_loader_mesh.argtypes = [vp, i32, C.POINTER(i32), C.c_char_p, C.c_char_p, i32, vp, vp]; L.ref_loader_mesh.restype = None
        L.ref_loader_material.argtypes = [vp, i32, C.c_char_p, C.c_char_p, i32, vp]; L.ref_loader_material.restype = None
        for name in ("ref_sizes", "ref_solve_quadratic", "ref_sphere_hit", "ref_sphere_normal_at", "ref_plane_hit", "ref_triangle_hit",
                     "ref_vec3_ops", "ref_camera_make", "ref_camera_drive", "ref_camera_rays", "ref_frame_spheres", "ref_frame_planes"):
            getattr(L, name).restype = None

    # ---- primitives: arrays in, arrays out -----------------------------------------------------------------------------
    def sizes(self):
        out = np.zeros(9, np.int32)
        self.lib.ref_sizes(_p(out))
        return dict(zip(("vec3", "Material", "Sphere", "Plane", "Triangle", "Camera", "Vertex", "Light", "Hit"), out.tolist()))

    def solve_quadratic(self, a, b, c):
        a, b, c = f32(a), f32(b), f32(c)
        n = len(a)
        ok = np.zeros(n, np.int32); x0 = np.zeros(n, np.float32); x1 = np.zeros(n, np.float32)
        self.lib.ref_solve_quadratic(n, _p(a), _p(b), _p(c), _p(ok), _p(x0), _p(x1))
        return ok, x0, x1

    def sphere_hit(self, centre, radius, origin, direction):
        centre, radius, origin, direction = f32(centre), f32(radius), f32(origin), f32(direction)
        n = len(radius)
        ok = np.zeros(n, np.int32); t0 = np.zeros(n, np.float32); t1 = np.zeros(n, np.float32)
        self.lib.ref_sphere_hit(n, _p(centre), _p(radius), _p(origin), _p(direction), _p(ok), _p(t0), _p(t1))
        return ok, t0, t1

    def plane_hit(self, p, normal, origin, direction):
        p, normal, origin, direction = f32(p), f32(normal), f32(origin), f32(direction)
        n = len(p)
        ok = np.zeros(n, np.int32); t0 = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
        self.lib.ref_plane_hit(n, _p(p), _p(normal), _p(origin), _p(direction), _p(ok), _p(t0), _p(nrm))
        return ok, t0, nrm

    def triangle_hit(self, a, b, c, origin, direction):
        a, b, c, origin, direction = f32(a), f32(b), f32(c), f32(origin), f32(direction)
        n = len(a)
        ok = np.zeros(n, np.int32); t0 = np.zeros(n, np.float32)
        self.lib.ref_triangle_hit(n, _p(a), _p(b), _p(c), _p(origin), _p(direction), _p(ok), _p(t0))
        return ok, t0

    def vec3_ops(self, a, b):
        a, b = f32(a), f32(b)
        n = len(a)
        out = [np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32),
               np.zeros((n, 3), np.float32), np.zeros(n, np.float32)]
        self.lib.ref_vec3_ops(n, _p(a), _p(b), *[_p(o) for o in out])
        return dict(zip(("length", "dot", "cross", "normalise", "reflect", "radians"), out))

    # ---- camera ------------------------------------------------------------------------------------------------------------
    def camera(self, position, fov, yaw, pitch, aspect=1.0):
        out = np.zeros(CAMERA_BYTES, np.uint8)
        pos = f32(position)
        self.lib.ref_camera_make(_p(pos), fov, aspect, yaw, pitch, _p(out))
        return out

    def camera_drive(self, cam84, moves=(), mouse=()):
        cam = np.array(cam84, np.uint8, copy=True)
        mv = np.ascontiguousarray(moves, np.int32); ms = np.ascontiguousarray(mouse, np.float64).reshape(-1, 2)
        self.lib.ref_camera_drive(_p(cam), _p(mv), len(mv), _p(ms), len(ms))
        return cam

    def camera_rays(self, cam84, width, height, window=None, offset=(0.5, 0.5)):
        x0, y0, x1, y1 = window if window else (0, 0, width, height)
        raw = np.zeros((y1 - y0, x1 - x0, 3), np.float32); direction = np.zeros_like(raw)
        cam = np.ascontiguousarray(cam84, np.uint8)
        self.lib.ref_camera_rays(_p(cam), width, height, x0, y0, x1, y1, offset[0], offset[1], _p(raw), _p(direction))
        return raw, direction

    def frame_spheres(self, cam84, width, height, centres, radii):
        centres, radii = f32(centres).reshape(-1, 3), f32(radii)
        n = len(radii)
        ok = np.zeros((height, width, n), np.uint8); t0 = np.zeros((height, width, n), np.float32)
        cam = np.ascontiguousarray(cam84, np.uint8)
        self.lib.ref_frame_spheres(_p(cam), width, height, n, _p(centres), _p(radii), _p(ok), _p(t0))
        return ok, t0

    def frame_planes(self, cam84, width, height, points, normals):
        points, normals = f32(points).reshape(-1, 3), f32(normals).reshape(-1, 3)
        n = len(points)
        ok = np.zeros((height, width, n), np.uint8); t0 = np.zeros((height, width, n), np.float32)
        cam = np.ascontiguousarray(cam84, np.uint8)
        self.lib.ref_frame_planes(_p(cam), width, height, n, _p(points), _p(normals), _p(ok), _p(t0))
        return ok, t0

    # ---- Model::hit without a box, through the program ref_model_kat ------------------------------------------------------
    @staticmethod
    def _run_model(mode, header, arrays, n):
        with tempfile.TemporaryDirectory() as tmp:
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as fh:
                fh.write(struct.pack("<4i", *header))
                for a in arrays:
                    fh.write(np.ascontiguousarray(a).tobytes())
            subprocess.check_call([REF_MODEL, mode, fin, fout])
            raw = open(fout, "rb").read()
        assert len(raw) == n * (4 + 4 + 36), (len(raw), n)
        hit = np.frombuffer(raw, np.int32, n, 0).copy()
        t0 = np.frombuffer(raw, np.float32, n, 4 * n).copy()
        attrs = np.frombuffer(raw, np.float32, 9 * n, 8 * n).reshape(n, 9).copy()
        return hit, t0, attrs

    def model_hit_nobox(self, verts12, indices, rays6, cull):
        """verts12 [v,12] position, normal, ambient, diffuse; indices int32; rays6 [r,6].  Returns hit, t0, attrs[r,9]."""
        verts12, rays6 = f32(verts12).reshape(-1, 12), f32(rays6).reshape(-1, 6)
        indices = np.ascontiguousarray(indices, np.int32)
        return self._run_model("mesh", (len(verts12), len(indices), len(rays6), int(bool(cull))), (verts12, indices, rays6), len(rays6))

    def triangle_cases(self, cases15, cull):
        """cases15 [n,15] v0, v1, v2, origin, direction: one single-triangle model per case.  attrs[:, 0:3] = (u, v, 1-u-v)."""
        cases15 = f32(cases15).reshape(-1, 15)
        return self._run_model("tri", (len(cases15), int(bool(cull)), 0, 0), (cases15,), len(cases15))

    # ---- objl::Loader --------------------------------------------------------------------------------------------------------
    def load_obj(self, path):
        """Same tuple as tests/test_loader_oracle.py builds from the host loader."""
        L = self.lib
        h = L.ref_loader_load(str(path).encode())
        try:
            counts = (C.c_int * 5)()
            L.ref_loader_counts(h, counts)
            ok, nv, ni, nm, nmat = tuple(counts)
            verts = np.zeros((max(nv, 1), 8), np.float32)
            idx = np.zeros(max(ni, 1), np.uint32)
            L.ref_loader_arrays(h, _p(verts), _p(idx))
            meshes = []
            for m in range(nm):
                c2 = (C.c_int * 2)()
                name = C.create_string_buffer(256); mname = C.create_string_buffer(256)
                mat = np.zeros(13, np.float32)
                L.ref_loader_mesh(h, m, c2, name, mname, 256, _p(mat), None)
                mv = np.zeros((max(c2[0], 1), 8), np.float32)
                L.ref_loader_mesh(h, m, c2, name, mname, 256, _p(mat), _p(mv))
                meshes.append((name.value.decode("latin-1"), c2[0], c2[1], mv[:c2[0]].copy(), mname.value.decode("latin-1"), mat.copy()))
            materials = []
            for k in range(nmat):
                name = C.create_string_buffer(256); mapkd = C.create_string_buffer(256)
                mat = np.zeros(13, np.float32)
                L.ref_loader_material(h, k, name, mapkd, 256, _p(mat))
                materials.append((name.value.decode("latin-1"), mapkd.value.decode("latin-1"), mat.copy()))
            return bool(ok), verts[:nv].copy(), idx[:ni].copy(), meshes, materials
        finally:
            L.ref_loader_free(h)


_ref = None


def ref():
    global _ref
    if _ref is None:
        _ref = Ref()
    return _ref
