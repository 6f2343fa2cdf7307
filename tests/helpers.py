"""Small scene-building helpers shared by the CPU and GPU tests."""
import numpy as np

from conftest import load_package

load_package()
from cuda_ray_tracer_amd import meshgen, scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402


def mesh_from_triangles(tris, normals=None, kd=None, ka=None, name="test"):
    """tris: [n, 3, 3] corner positions (OBJ space).  Optional per-corner normals / Kd / Ka of the same shape."""
    tris = np.asarray(tris, np.float32)
    n = tris.shape[0] * 3
    nrm = np.asarray(normals, np.float32).reshape(n, 3) if normals is not None else np.tile(np.float32([0, 0, 1]), (n, 1))
    kd_ = np.asarray(kd, np.float32).reshape(n, 3) if kd is not None else np.tile(np.float32([0.5, 0.5, 0.5]), (n, 1))
    ka_ = np.asarray(ka, np.float32).reshape(n, 3) if ka is not None else np.ones((n, 3), np.float32)
    return meshgen.LoadedMesh(name, tris.reshape(n, 3), nrm, np.zeros((n, 2), np.float32), ka_, kd_,
                              np.full((n, 3), 0.5, np.float32), np.arange(n, dtype=np.uint32), [("m", 0, tris.shape[0])])


def simple_spec(**kw):
    """A scene with one light and nothing else unless given."""
    s = scenes.SceneSpec(light=(0.0, -50.0, 0.0))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


MATTE = ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 0.3, 0.7, 0.0, 10.0)
