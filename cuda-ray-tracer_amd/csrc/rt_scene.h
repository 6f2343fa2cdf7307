// Host-side flattening of an rt_scene_desc into the device layout of rt_device_types.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_device_types.h"
#include "rt_types.h"

namespace rt {

struct OctreeParams {
    int maxDepth = 8;        // levels of internal nodes
    bool widen = true;       // fill free child slots of a node with its grandchildren (host builder)
    bool optimalCollapse = true; // host builder, with sah: the binary nodes that stay nodes are chosen so that their total surface area is least (RT_COLLAPSE=0: open the largest child while there is room)
    bool sah = true;         // host builder: binary surface-area splits folded into eight-wide nodes, instead of octant splits
    int sahLeafTriangles = 2; // leaf size of the surface-area builder when the caller names none (measured, config 3: 1 -> +9 %, 3 -> +0.4 %, 4 -> +1.5 %)
    int leafTriangles = 6;   // a node with at most this many triangles is a leaf (measured on MI355X, config 3: 4 -> +2 %, 8 -> +2 %, 16 -> +18 % frame time)
};

// What the device octree builder needs to know about one model with reachable triangles.
struct DeviceBuildModel {
    int32_t model;
    uint32_t rootIndex;         // node reserved for the model's root
    uint32_t triStart, triCount;   // the model's slice of the scene-wide triangle numbering (triStart = triBase)
    float lo[3], hi[3];         // bounds of the reachable triangles
    float pad;                  // padding of this model's child boxes
    int32_t perBoxTris;         // triangles per sequential chunk (eight-box mode)
    int32_t chunkIndices[8];    // boundingBoxes[b].indicesCount
};

// Host staging copy of everything that goes to HBM.
struct FlatScene {
    std::vector<F4> sphGeom, sphMat, plnP, plnN, plnMat, plnNormals, boxMin, boxMax, boxMat;
    std::vector<F4> mdlGateMin, mdlGateMax, mdlTightMin, mdlTightMax, mdlMat, mdlChunkMin, mdlChunkMax;
    bool chunksSequential = true;   // every model's chunk boxes list its triangles in file order (needed by the eight-box mode)
    std::vector<uint32_t> mdlInfo;
    std::vector<F4> nodes, tris, attrs;
    float lightPos[3] = {0, 0, 0}, lightCol[3] = {0, 0, 0};
    int64_t triangleCount = 0, nodeCount = 0, leafCount = 0;
    int maxDepth = 0;               // deepest nesting of internal nodes, forest levels included
    uint32_t forestRoot = kNoRoot;  // node the mesh traversal starts from
    int forestDepth = 0;            // nesting levels above the model roots
    OctreeParams params;            // the build parameters in force
    std::vector<DeviceBuildModel> deviceModels;   // filled instead of the per-model octrees when meshes are built on the GPU
    int prebuiltOctrees = 0;        // models whose octree the caller supplied (rt_model.octree); such a scene is validated at upload
};

// Number of triangles the reference's loops visit for a model with this many indices
// (for (i = 0; i < indicesCount - 3; i += 3), Model.cpp:282): the last triangle is never reached.
int64_t reachableTriangles(int64_t indicesCount);

// Fills `out` from `scene`.  Returns false and sets `err` if the scene fails validation.  With meshesOnDevice the
// per-model octrees, triangle and attribute records are left out (nodes then holds only the forest above the model
// roots and the reserved root slots) and deviceModels describes what the device builder has to produce.
bool flattenScene(const rt_scene_desc& scene, FlatScene& out, std::string& err, bool meshesOnDevice = false);

// Walks the flattened octrees of `flat` (built from `scene`) and checks the invariants the traversal kernel relies on:
// the forest reaches every non-empty model's root exactly once; every reachable triangle of every model sits in
// exactly one leaf below that root, inside the padded box of the slot that holds it and tagged with its model;
// child boxes nest; internal children are contiguous and in range; no path is deeper than kMaxOctreeDepth.
bool validateFlatScene(const rt_scene_desc& scene, const FlatScene& flat, std::string& err);

// Packs one material into the three-float4 device form.
void packMaterial(const rt_material& m, int objectType, uint32_t flags, F4 out[3]);

}  // namespace rt
