// Host mirror of Model (Model.h:10-44, Model.cpp:29-129): loads an OBJ through objl::Loader and bakes it into world
// space with its bounding volumes.  The baking itself is the library's rt_bake_model, which restates the body of
// the reference constructor (integer size, mirrored Y, origin-anchored padded boxes, last triangle never copied).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "AABB.h"
#include "Object.h"
#include "Octree.h"
#include "rt_api.h"

class Model : public Object {
public:
    Model() { objectName = Model_t; }
    Model(vec3 position, int size, const char* filename, Material mat, bool debug = false, ObjectType objectType = Diffuse);

    vec3 position;      // shadows Object::position, as in the reference (Model.h:27)
    int size = 1;

    AABB boundingBox;
    int bbCount = 8;
    AABB boundingBoxes[8];

    std::vector<Vertex> vertices;
    int vertCount = 0;
    int materialCount = 0;
    int meshCount = 0;
    std::vector<int> indices;
    int indicesCount = 0;

    bool loaded = false;            // false if the OBJ file could not be read (the model is then empty)

    // Model.h:39 (Octree octree): built by the constructor over the baked triangles.  It holds the flat node and triangle
    // arrays the kernels read; toRt() passes them on, so rt_upload_scene uploads this very tree.  Shared by copies of the
    // model (the arrays are never modified after the build).
    std::shared_ptr<Octree> octree;

    // A separate octree over a triangle soup of the reachable triangles, with the OctreeNode pointer view expanded
    // (for host code that wants to walk nodes; soupStorage must outlive the tree).
    void buildOctree(Octree& octree, std::vector<Vertex>& soupStorage) const;

    // View of this model for rt_upload_scene; valid while the Model is alive and unmodified.
    rt_model toRt() const;
};
