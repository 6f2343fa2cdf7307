"""ctypes binding of the renderer's C ABI (include/rt_api.h).

The shared library is built in-tree by csrc/Makefile.  There is no CPU fallback: if the library is missing or
the GPU call fails, the error is raised.
"""
import ctypes as C
import os

import numpy as np

from . import ctypes_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB_OVERRIDE") or os.path.join(_HERE, "csrc", "librt_amd.so")

RT_TILE = 32
_lib = None


class RtError(RuntimeError):
    pass


def lib():
    """Loads librt_amd.so (once).  torch, when used in the same process, must be imported first so that both
    share one HIP runtime; the caller decides that, this module never imports torch."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}`")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.rt_last_error.restype = C.c_char_p
    L.rt_create.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    L.rt_destroy.argtypes = [vp]
    L.rt_destroy.restype = None
    L.rt_upload_scene.argtypes = [vp, C.POINTER(T.SceneDesc)]
    L.rt_update_frame_state.argtypes = [vp, C.POINTER(T.Camera), C.POINTER(T.Config), C.POINTER(T.Dynamic)]
    L.rt_set_strict_octree_flag.argtypes = [vp, i32]
    L.rt_update_objects.argtypes = [vp, C.POINTER(T.SceneDesc)]
    L.rt_render_frame.argtypes = [vp, i32, i32, vp, C.POINTER(T.FrameStats)]
    L.rt_render_frame_ids.argtypes = [vp, i32, i32, vp, vp, vp, C.POINTER(T.FrameStats)]
    L.rt_render_device.argtypes = [vp, i32, i32, vp, vp]
    L.rt_render_device_static.argtypes = [vp, i32, i32, vp, vp]
    L.rt_render_path_device.argtypes = [vp, i32, i32, i32, C.POINTER(T.Camera), vp, vp]
    L.rt_render_path_tiles_device.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(T.Camera), vp, sz, vp]
    L.rt_tile_buffer_bytes.argtypes = [i32, i32, i32]
    L.rt_tile_buffer_bytes.restype = sz
    L.rt_tile_buffer_bytes_dealt.argtypes = [i32, i32, i32, i32, i32]
    L.rt_tile_buffer_bytes_dealt.restype = sz
    L.rt_set_tile_deal.argtypes = [vp, i32, i32]
    L.rt_get_tile_deal.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.rt_render_tiles_device.argtypes = [vp, i32, i32, i32, i32, vp, sz, vp]
    L.rt_untile_device.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.rt_untile_strided_device.argtypes = [vp, i32, i32, i32, vp, sz, vp, vp]
    L.rt_untile_batch_device.argtypes = [vp, i32, i32, i32, vp, sz, sz, i32, vp, vp]
    L.rt_count_rays.argtypes = [vp, i32, i32, C.POINTER(T.FrameStats)]
    L.rt_get_scene_info.argtypes = [vp, C.POINTER(T.SceneInfo)]
    L.rt_profile_waves.argtypes = [vp, i32, i32, vp, sz, C.POINTER(sz)]
    L.rt_validate_scene_host.argtypes = [C.POINTER(T.SceneDesc), C.POINTER(T.SceneInfo)]
    L.rt_build_octree.argtypes = [C.POINTER(T.Model), i32, i32, C.POINTER(T.FlatOctree)]
    L.rt_free_octree.argtypes = [C.POINTER(T.FlatOctree)]
    L.rt_free_octree.restype = None
    L.rt_validate_scene_device.argtypes = [C.c_void_p, C.POINTER(T.SceneDesc)]
    L.rt_camera_look.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.POINTER(T.Camera)]
    L.rt_bake_model.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, C.POINTER(C.c_float), i32,
                                vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RtError(f"rt error {rc}: {lib().rt_last_error().decode()}")


def camera_look(position, fov=90.0, yaw=-90.0, pitch=0.0):
    """Camera basis exactly as the host library computes it (Camera.cpp:5-20, 54-63)."""
    cam = T.Camera()
    pos = (C.c_float * 3)(*[float(v) for v in position])
    _check(lib().rt_camera_look(pos, float(fov), float(yaw), float(pitch), C.byref(cam)))
    return cam


class Renderer:
    """One GPU context: upload a scene, set the frame state, render."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        dev = (C.c_int * 1)(int(device))
        _check(lib().rt_create(1, dev, C.byref(self._ctx)))

    def close(self):
        if self._ctx:
            lib().rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene_desc):
        _check(lib().rt_upload_scene(self._ctx, C.byref(scene_desc)))

    def update_objects(self, scene_desc):
        """Replaces spheres, planes, boxes and the light; the uploaded models stay (rt_update_objects)."""
        _check(lib().rt_update_objects(self._ctx, C.byref(scene_desc)))

    def set_state(self, camera, config, dynamic=None):
        _check(lib().rt_update_frame_state(self._ctx, C.byref(camera), C.byref(config),
                                           C.byref(dynamic) if dynamic is not None else None))

    def set_strict_octree_flag(self, strict):
        _check(lib().rt_set_strict_octree_flag(self._ctx, 1 if strict else 0))

    def render(self, width, height):
        """Returns (rgb uint8 [h, w, 3], FrameStats)."""
        rgb = np.empty((height, width, 3), dtype=np.uint8)
        st = T.FrameStats()
        _check(lib().rt_render_frame(self._ctx, width, height, rgb.ctypes.data, C.byref(st)))
        return rgb, st

    def render_ids(self, width, height):
        """Returns (rgb, ids int32 [h, w, 2], t float32 [h, w], FrameStats) from the instrumented kernel."""
        rgb = np.empty((height, width, 3), dtype=np.uint8)
        ids = np.empty((height, width, 2), dtype=np.int32)
        t = np.empty((height, width), dtype=np.float32)
        st = T.FrameStats()
        _check(lib().rt_render_frame_ids(self._ctx, width, height, rgb.ctypes.data, ids.ctypes.data, t.ctypes.data, C.byref(st)))
        return rgb, ids, t, st

    def count_rays(self, width, height):
        st = T.FrameStats()
        _check(lib().rt_count_rays(self._ctx, width, height, C.byref(st)))
        return st

    def render_device(self, width, height, d_rgb_ptr, stream=None):
        _check(lib().rt_render_device(self._ctx, width, height, C.c_void_p(d_rgb_ptr), C.c_void_p(stream or 0)))

    def render_device_static(self, width, height, d_rgb_ptr, stream=None):
        """rt_render_device with the dispatch feedback in the library's never-allocated device array."""
        _check(lib().rt_render_device_static(self._ctx, width, height, C.c_void_p(d_rgb_ptr), C.c_void_p(stream or 0)))

    def render_path_device(self, width, height, cameras, d_rgb_frames_ptr, stream=None):
        """One launch for len(cameras) frames; frame f goes to d_rgb_frames_ptr + f * 3 * width * height."""
        arr = (T.Camera * len(cameras))(*cameras)
        _check(lib().rt_render_path_device(self._ctx, width, height, len(cameras), arr, C.c_void_p(d_rgb_frames_ptr), C.c_void_p(stream or 0)))

    def render_path_tiles_device(self, width, height, rank, world, cameras, d_tiles_ptr, frame_stride_bytes, stream=None):
        arr = (T.Camera * len(cameras))(*cameras)
        _check(lib().rt_render_path_tiles_device(self._ctx, width, height, rank, world, len(cameras), arr, C.c_void_p(d_tiles_ptr),
                                                 frame_stride_bytes, C.c_void_p(stream or 0)))

    def set_tile_deal(self, rank0_slots, other_slots):
        """How the tile entry points of this context deal a frame's tiles to the ranks (rt_set_tile_deal; default 1 / 1)."""
        _check(lib().rt_set_tile_deal(self._ctx, rank0_slots, other_slots))

    def render_tiles_device(self, width, height, rank, world, d_tiles_ptr, capacity, stream=None):
        _check(lib().rt_render_tiles_device(self._ctx, width, height, rank, world, C.c_void_p(d_tiles_ptr), capacity,
                                            C.c_void_p(stream or 0)))

    def untile_device(self, width, height, world, d_gathered_ptr, d_rgb_ptr, stream=None):
        _check(lib().rt_untile_device(self._ctx, width, height, world, C.c_void_p(d_gathered_ptr), C.c_void_p(d_rgb_ptr),
                                      C.c_void_p(stream or 0)))

    def untile_strided_device(self, width, height, world, d_gathered_ptr, rank_stride_bytes, d_rgb_ptr, stream=None):
        _check(lib().rt_untile_strided_device(self._ctx, width, height, world, C.c_void_p(d_gathered_ptr), rank_stride_bytes,
                                              C.c_void_p(d_rgb_ptr), C.c_void_p(stream or 0)))

    def untile_batch_device(self, width, height, world, d_gathered_ptr, rank_stride_bytes, frame_stride_bytes, n_frames, d_rgb_frames_ptr, stream=None):
        _check(lib().rt_untile_batch_device(self._ctx, width, height, world, C.c_void_p(d_gathered_ptr), rank_stride_bytes, frame_stride_bytes,
                                            n_frames, C.c_void_p(d_rgb_frames_ptr), C.c_void_p(stream or 0)))

    def profile_waves(self, width, height):
        """Per-wave diagnostic records of one instrumented frame: uint64 array [waves, 18] (see rt_profile_waves)."""
        tiles = ((width + RT_TILE - 1) // RT_TILE) * ((height + RT_TILE - 1) // RT_TILE)
        cap = (tiles + 8) * (RT_TILE // 8) ** 2 + 4096    # + split waves of the heaviest blocks
        buf = np.zeros((cap, 18), np.uint64)
        n = C.c_size_t(0)
        _check(lib().rt_profile_waves(self._ctx, width, height, buf.ctypes.data, cap, C.byref(n)))
        return buf[:n.value]

    def validate_device_scene(self, scene_desc):
        """Checks the octree arrays resident on the GPU against the scene they were uploaded from (raises on violation)."""
        _check(lib().rt_validate_scene_device(self._ctx, C.byref(scene_desc)))

    def scene_info(self):
        info = T.SceneInfo()
        _check(lib().rt_get_scene_info(self._ctx, C.byref(info)))
        return info


def validate_scene_host(scene_desc):
    """Flattens the scene on the host and checks the octree invariants; returns SceneInfo (raises on violation)."""
    info = T.SceneInfo()
    _check(lib().rt_validate_scene_host(C.byref(scene_desc), C.byref(info)))
    return info


class FlatOctree:
    """A model's octree as the flat node / triangle arrays the kernels read (rt_build_octree); freed with the object."""

    def __init__(self, model, max_depth=0, leaf_triangles=0):
        self.c = T.FlatOctree()
        _check(lib().rt_build_octree(C.byref(model), max_depth, leaf_triangles, C.byref(self.c)))

    def nodes(self):
        """uint32 view [nodeCount, 32] of the node records."""
        n = int(self.c.nodeCount)
        return np.ctypeslib.as_array(C.cast(self.c.nodes, C.POINTER(C.c_uint32)), shape=(n, 32)) if n else np.zeros((0, 32), np.uint32)

    def triangles(self):
        """uint32 view [triangleRecords, 12] of the triangle records."""
        n = int(self.c.triangleRecords)
        return np.ctypeslib.as_array(C.cast(self.c.triangles, C.POINTER(C.c_uint32)), shape=(n, 12)) if n else np.zeros((0, 12), np.uint32)

    def __del__(self):
        try:
            lib().rt_free_octree(C.byref(self.c))
        except Exception:
            pass


def tile_buffer_bytes(width, height, world, deal=(1, 1)):
    return int(lib().rt_tile_buffer_bytes_dealt(width, height, world, deal[0], deal[1]))
