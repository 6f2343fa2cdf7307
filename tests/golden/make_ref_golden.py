#!/usr/bin/env python3
"""Writes tests/golden/refgen_*.npz: fixtures whose expected values were computed by THE REFERENCE'S OWN compiled routines
(oracle/_ref, built from the reference checkout by oracle/build_ref.sh), not by the oracle restatement.

Needs the reference checkout (this container has it; the GPU box does not, which is why the outputs are committed):
    python tests/golden/make_ref_golden.py

Analytic frames (refgen_scene<k>_<w>x<h>.npz): the spheres and planes of reference scene k (RayTracer.cpp:98-124, 127-157,
229-254) seen through that scene's camera.  For every pixel the reference's Camera::rasterToCameraSpace + normalise give the
primary ray (kernels.cu:355-356) and its Sphere::hit / Plane::hit give (hit, t0) per object; this script only reduces them
with the rule of kernels.cu:81 (`t0 < hit.t`, hit.t starting at 999, spheres before planes, lowest index first):
    t   [h, w] float32  closest distance, 999 where nothing is hit
    id  [h, w] int32    objectName << 24 | index (the identity rt_render_frame_ids reports), -1 where nothing is hit
    inputs: cam_pos, fov, yaw, pitch, sphere centres / radii, plane points / normals

Ray fixture (refgen_rays.npz): cameras and the reference's raw and normalised ray directions on small non-square frames, pixel
centres and the four anti-aliasing offsets.

Mesh fixtures (refgen_mesh_<name>.npz): Model::hit without a box (Model.cpp:318-356, with Model::triangleIntersect
Model.cpp:135-159) for the primary rays of a small frame on a generated, baked mesh: hit, t0 and the interpolated normal /
Ka / Kd per pixel; inputs: the baked vertex arrays and indices, camera, cull flag.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import scenes  # noqa: E402
from cuda_ray_tracer_amd import ctypes_types as T  # noqa: E402
import ref_lib  # noqa: E402

f32 = np.float32
FAR = f32(999.0)

ANALYTIC = [(1, 640, 480), (1, 400, 150), (2, 640, 480), (2, 400, 150), (5, 640, 480), (5, 150, 400)]
MESHES = [  # name, detail, position, size, camera (pos, yaw, pitch), frame, cull
    ("Chalice", 0.15, (0.0, 6.0, 10.0), 10, ((-11.2339, -4.81805, -13.431), -311.2, 13.8), (160, 90), 1),
    ("cube", 1.0, (0.0, 0.0, -12.0), 4, ((0.5, 0.25, -12.5), -60.0, 10.0), (120, 50), 0),     # camera inside: hits with t < 0 win
    ("Table", 0.1, (0.0, 19.0, 0.0), 15, ((-11.2339, -4.81805, -13.431), -311.2, 13.8), (128, 72), 1),
]


def closest(sph, pln):
    """kernels.cu:81 over spheres then planes: an object replaces the current hit iff it is hit and t0 < hit.t."""
    (ok_s, t_s), (ok_p, t_p) = sph, pln
    h, w = ok_s.shape[:2]
    t = np.full((h, w), FAR, np.float32); ident = np.full((h, w), -1, np.int32)
    for kind, ok, t0 in ((T.RT_SPHERE_T, ok_s, t_s), (T.RT_PLANE_T, ok_p, t_p)):
        for i in range(ok.shape[2]):
            better = (ok[..., i] == 1) & (t0[..., i] < t)
            t[better] = t0[..., i][better]
            ident[better] = (kind << 24) | i
    return t, ident


def analytic(R, index, w, h):
    spec = scenes.reference_scene(index)
    cfg = scenes.default_config()
    scenes.apply_frame_fixups(spec, cfg)          # spheres[0].radius = softShadowRadius (Window.cpp:382)
    cam = R.camera(spec.cam_pos, spec.fov, spec.cam_yaw, spec.cam_pitch)
    centres = f32([s[0] for s in spec.spheres]); radii = f32([s[1] for s in spec.spheres])
    points = f32([p[0] for p in spec.planes]); normals = f32([p[1] for p in spec.planes])
    t, ident = closest(R.frame_spheres(cam, w, h, centres, radii), R.frame_planes(cam, w, h, points, normals))
    return dict(t=t, id=ident, scene=np.int32(index), width=np.int32(w), height=np.int32(h), cam_pos=f32(spec.cam_pos),
                fov=f32(spec.fov), yaw=f32(spec.cam_yaw), pitch=f32(spec.cam_pitch), centres=centres, radii=radii,
                points=points, normals=normals)


def rays(R):
    rng = np.random.default_rng(5)
    out = {}
    cams = [((0, 0, 0), 90.0, -90.0, 0.0), ((1.75159, -55.2005, 144.58), 90.0, -93.5, 6.6), ((-11.2339, -4.81805, -13.431), 90.0, -311.2, 13.8),
            ((3.0, -2.0, 7.0), 37.5, 412.25, -89.0), (tuple(rng.normal(size=3) * 40), 120.0, 17.0, 55.5)]
    sizes = [(64, 24), (24, 64), (33, 17), (16, 16), (7, 50)]
    offsets = [(0.5, 0.5), (0.25, 0.25), (0.25, 0.75), (0.75, 0.25), (0.75, 0.75)]
    out["cams"] = f32([[*p, fov, yaw, pitch] for p, fov, yaw, pitch in cams])
    out["sizes"] = np.int32(sizes); out["offsets"] = np.float64(offsets)
    for i, ((p, fov, yaw, pitch), (w, h)) in enumerate(zip(cams, sizes)):
        cam = R.camera(p, fov, yaw, pitch)
        fields = ref_lib.camera_fields(cam)
        out[f"basis{i}"] = np.concatenate([fields["position"], fields["up"], fields["right"], fields["direction"], [fields["fovFactor"]]]).astype(np.float32)
        for j, off in enumerate(offsets):
            raw, direction = R.camera_rays(cam, w, h, offset=off)
            out[f"raw{i}_{j}"] = raw; out[f"dir{i}_{j}"] = direction
    return out


def mesh_arrays(name, detail, position, size):
    """The baked vertex arrays the renderer is given for this model: [v, 12] position, normal, ambient, diffuse; indices."""
    spec = scenes.SceneSpec(light=(0.0, -50.0, 0.0))
    spec.models = [scenes.ModelSpec(name, position, size, ((1, 1, 1), (1, 1, 1), 1.0, 0.0, 0.0, 10.0), False, T.RT_DIFFUSE, detail)]
    built = scenes.build_scene(spec)
    model = built.desc.models[0]
    verts = np.ctypeslib.as_array(C.cast(model.vertices, C.POINTER(C.c_float)), shape=(model.vertCount, 19)).copy()
    verts12 = np.concatenate([verts[:, 0:3], verts[:, 6:9], verts[:, 10:13], verts[:, 13:16]], axis=1).astype(np.float32)
    indices = np.ctypeslib.as_array(model.indices, shape=(model.indicesCount,)).copy()
    return verts12, indices


def mesh(R, name, detail, position, size, camera, frame, cull):
    verts12, indices = mesh_arrays(name, detail, position, size)
    (cam_pos, yaw, pitch), (w, h) = camera, frame
    cam = R.camera(cam_pos, 90.0, yaw, pitch)
    _, dirs = R.camera_rays(cam, w, h)
    rays6 = np.concatenate([np.tile(f32(cam_pos), (w * h, 1)), dirs.reshape(-1, 3)], axis=1)
    hit, t0, attrs = R.model_hit_nobox(verts12, indices, rays6, cull)
    return dict(hit=hit.reshape(h, w).astype(np.uint8), t0=t0.reshape(h, w), attrs=attrs.reshape(h, w, 9), verts12=verts12,
                indices=indices, name=np.str_(name), detail=f32(detail), position=f32(position), size=np.int32(size), cam_pos=f32(cam_pos),
                yaw=f32(yaw), pitch=f32(pitch), width=np.int32(w), height=np.int32(h), cull=np.int32(cull))


def fixtures(R):
    """Yields (file name, dict of arrays) for every fixture."""
    for index, w, h in ANALYTIC:
        yield f"refgen_scene{index}_{w}x{h}.npz", analytic(R, index, w, h)
    yield "refgen_rays.npz", rays(R)
    for args in MESHES:
        yield f"refgen_mesh_{args[0]}.npz", mesh(R, *args)


def main():
    if not ref_lib.available():
        sys.exit("oracle/_ref cannot be built here (no reference checkout): fixtures left as they are")
    for name, arrays in fixtures(ref_lib.ref()):
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **arrays)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
