"""MI355X-native renderer for the reference ray tracer's per-pixel hot path.

The directory name contains a hyphen, so the package is loaded by path (see `load_package` in the
repository's tests/conftest.py and bench.py) under the module name ``cuda_ray_tracer_amd``.

Layout
  csrc/       gfx950 HIP kernels, scene flattening / octree build, and the C ABI (librt_amd.so)
  host/       C++ mirror of the reference's scene API (Camera, Sphere, Plane, AABB, Model, RayTracer, ...)
  binding.py  ctypes binding of include/rt_api.h
  scenes.py   the reference's scene tables and key presets as data
  meshgen.py  deterministic procedural stand-ins for the reference's (absent) OBJ meshes
"""
from . import ctypes_types  # noqa: F401
