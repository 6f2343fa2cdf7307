#!/usr/bin/env python3
"""Scene-ingest throughput at the scale of benchmark configuration 5: OBJ text -> LoadedVertices / LoadedIndices -> baked model
-> device scene.

  python tools/loader_throughput.py [detail] [--upload]

For the two meshes of configuration 5 (Chicken, WoodLog) at the given mesh detail (7 = the 10.3 M-triangle scene), written once
as OBJ text: the host loader (cuda-ray-tracer_amd/host/OBJ_Loader.cpp) and - where oracle/_ref exists - the reference's own
objl::Loader (OBJ_Loader.cpp:183-465) parse the same files; then the product's bake (rt_bake_model, Model.cpp:62-126) of one
copy, and with --upload (needs the GPU) the upload of all 144 copies with the octrees built on the host and on the GPU.
The reference's Model constructor additionally prints one line per vertex (Model.cpp:114), which is not timed here.
Prints one JSON line.
"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

load_package()
from cuda_ray_tracer_amd import meshgen, scenes  # noqa: E402
import host_lib  # noqa: E402
import ref_lib  # noqa: E402


def best_of(fn, n=3):
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def main():
    detail = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 7.0
    upload = "--upload" in sys.argv
    L = host_lib.lib()
    out = {"detail": detail, "files": {}}
    with tempfile.TemporaryDirectory() as tmp:
        meshes = host_lib.write_models(tmp, ["Chicken", "WoodLog"], detail)
        total_tris = total_bytes = 0
        total_host = total_ref = 0.0
        for name, mesh in meshes.items():
            path = os.path.join(tmp, name + ".obj")
            size = os.path.getsize(path)
            tris = mesh.triangle_count

            def host_load():
                h = L.rth_loader_load(path.encode())
                counts = (C.c_int * 5)()
                L.rth_loader_counts(h, counts)
                L.rth_loader_free(h)
                return tuple(counts)

            t_host, counts = best_of(host_load)
            assert counts[0] == 1 and counts[2] == 3 * tris, counts
            row = {"obj_bytes": size, "triangles": tris, "host_loader_s": t_host, "host_loader_MB_per_s": size / t_host / 1e6,
                   "host_loader_Mtri_per_s": tris / t_host / 1e6}
            if ref_lib.available():
                R = ref_lib.ref()

                def ref_load():
                    h = R.lib.ref_loader_load(path.encode())
                    counts = (C.c_int * 5)()
                    R.lib.ref_loader_counts(h, counts)
                    R.lib.ref_loader_free(h)
                    return tuple(counts)

                t_ref, rcounts = best_of(ref_load, 2)
                assert rcounts == counts, (rcounts, counts)
                row.update({"reference_loader_s": t_ref, "reference_loader_MB_per_s": size / t_ref / 1e6,
                            "reference_loader_Mtri_per_s": tris / t_ref / 1e6, "speedup": t_ref / t_host})
                total_ref += t_ref
            t_bake, _ = best_of(lambda: scenes.bake_model(mesh, (0.0, 20.0, 0.0), 14))
            row["bake_s"] = t_bake
            out["files"][name] = row
            total_tris += tris; total_bytes += size; total_host += t_host
        # configuration 5 loads each of the two files 72 times (the reference has no instancing: every model is a parse and a bake)
        out["config5_ingest_estimate"] = {"models": 144, "triangles": 72 * total_tris, "obj_bytes_parsed": 72 * total_bytes,
                                          "host_loader_s": 72 * total_host,
                                          "reference_loader_s": 72 * total_ref if total_ref else None,
                                          "bake_s": 72 * sum(r["bake_s"] for r in out["files"].values())}
    if upload:
        from cuda_ray_tracer_amd import binding
        spec, cfg, w, h = scenes.config_scene(5, detail=detail)
        t0 = time.perf_counter()
        built = scenes.build_scene(spec)
        out["bake_all_models_s"] = time.perf_counter() - t0
        r = binding.Renderer(0)
        for on_device in (False, True):
            built.desc.buildOnDevice = 1 if on_device else 0
            t0 = time.perf_counter()
            r.upload(built.desc)
            out["upload_octrees_on_" + ("gpu" if on_device else "host") + "_s"] = time.perf_counter() - t0
        out["triangles_uploaded"] = int(r.scene_info().triangleCount)
        r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
