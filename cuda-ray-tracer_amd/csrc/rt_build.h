// Octree construction on the GPU (rt_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "rt_scene.h"

namespace rt {

// Device arrays produced by the builder; owned by the caller (hipFree / freeDeviceMeshes).
struct DeviceMeshes {
    F4* nodes = nullptr;       // forest nodes copied from the host staging copy + the per-model octrees
    F4* tris = nullptr;        // leaf-ordered triangle records, one spare record at the end
    F4* attrs = nullptr;       // shading attributes in scene-wide triangle order
    int64_t nodeCount = 0, leafCount = 0, triangleRecords = 0;
    int depth = 0;             // deepest nesting of internal nodes inside a model
};

// Builds the mesh arrays for `flat` (made by flattenScene(..., meshesOnDevice = true) from `scene`) on the current
// device.  Blocking; all temporaries are released before it returns.
hipError_t buildMeshesOnDevice(const rt_scene_desc& scene, const FlatScene& flat, DeviceMeshes& out, std::string& err);

void freeDeviceMeshes(DeviceMeshes& m);

}  // namespace rt
