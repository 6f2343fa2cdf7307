"""ctypes access to the C++ host mirror (cuda-ray-tracer_amd/host/librt_host.so) for tests."""
import ctypes as C
import os
import shutil

from conftest import ROOT, load_package

load_package()
from cuda_ray_tracer_amd import binding, meshgen  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402

HOST_SO = os.path.join(ROOT, "cuda-ray-tracer_amd", "host", "librt_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        binding.lib()       # librt_amd first (RTLD_GLOBAL), the host library links against it
        L = C.CDLL(HOST_SO)
        vp, i32 = C.c_void_p, C.c_int
        L.rth_create.argtypes = [i32, i32, C.c_char_p, i32]; L.rth_create.restype = vp
        L.rth_destroy.argtypes = [vp]; L.rth_destroy.restype = None
        L.rth_key_preset.argtypes = [vp, C.c_char]; L.rth_key_held.argtypes = [vp, C.c_char]; L.rth_key_held.restype = None
        L.rth_display.argtypes = [vp]; L.rth_display.restype = None
        L.rth_framebuffer.argtypes = [vp]; L.rth_framebuffer.restype = C.POINTER(C.c_ubyte)
        L.rth_last_rays.argtypes = [vp]; L.rth_last_rays.restype = C.c_ulonglong
        L.rth_add_sphere.argtypes = [vp, C.POINTER(C.c_float), C.c_float, C.POINTER(T.Material), i32]; L.rth_add_sphere.restype = None
        L.rth_raytrace_reference.argtypes = [vp, vp]
        L.rth_raytrace_shutdown.argtypes = []; L.rth_raytrace_shutdown.restype = None
        L.rth_describe.argtypes = [vp]; L.rth_describe.restype = C.POINTER(T.SceneDesc)
        L.rth_camera.argtypes = [vp, C.POINTER(T.Camera)]; L.rth_camera.restype = None
        L.rth_config.argtypes = [vp, C.POINTER(T.Config)]; L.rth_config.restype = None
        L.rth_model_load.argtypes = [C.POINTER(C.c_float), i32, C.c_char_p, C.POINTER(T.Material), i32, i32]; L.rth_model_load.restype = vp
        L.rth_model_free.argtypes = [vp]; L.rth_model_free.restype = None
        L.rth_model_loaded.argtypes = [vp]
        L.rth_model_view.argtypes = [vp, C.POINTER(T.Model)]; L.rth_model_view.restype = None
        L.rth_model_counts.argtypes = [vp, C.POINTER(i32)]; L.rth_model_counts.restype = None
        L.rth_model_octree_stats.argtypes = [vp, i32, i32, C.POINTER(i32)]; L.rth_model_octree_stats.restype = None
        L.rth_loader_load.argtypes = [C.c_char_p]; L.rth_loader_load.restype = vp
        L.rth_loader_free.argtypes = [vp]; L.rth_loader_free.restype = None
        L.rth_loader_counts.argtypes = [vp, C.POINTER(i32)]; L.rth_loader_counts.restype = None
        L.rth_loader_arrays.argtypes = [vp, vp, vp]; L.rth_loader_arrays.restype = None
        L.rth_loader_mesh.argtypes = [vp, i32, C.POINTER(i32), C.c_char_p, C.c_char_p, i32, vp, vp]; L.rth_loader_mesh.restype = None
        L.rth_loader_material.argtypes = [vp, i32, C.c_char_p, C.c_char_p, i32, vp]; L.rth_loader_material.restype = None
        L.rth_triangulate.argtypes = [C.POINTER(C.c_float), i32, C.POINTER(C.c_uint), i32]
        _lib = L
    return _lib


def write_models(directory, names, detail):
    """Writes <name>.obj (synthetic) and <name>.mtl (the reference's own) for each model into `directory`."""
    os.makedirs(directory, exist_ok=True)
    meshes = {}
    for name in names:
        mesh = meshgen.generate(name, detail)
        meshgen.write_obj(mesh, os.path.join(directory, name + ".obj"))
        shutil.copy(os.path.join(meshgen.ASSET_DIR, name + ".mtl"), os.path.join(directory, name + ".mtl"))
        meshes[name] = mesh
    return meshes
