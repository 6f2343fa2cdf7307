// Host mirror of Octree (Octree.h:6-17) with the reference's interface - init(vertices, count), build(node), root - and a
// different product: build() emits the flattened, cache-line-aligned node array plus the leaf-ordered triangle records
// that the gfx950 traversal kernel reads (rt_flat_octree, built by the renderer library's own builder through
// rt_build_octree), and a Model hands exactly these arrays to rt_upload_scene.  The reference's as-written build
// (Octree.cpp:11-43) deals the vertex list out sequentially and is never called, so only its interface is kept; the split
// is spatial.  The pointer-linked form of the reference (OctreeNode: box, leaf flag, eight contiguous children, vertices)
// is still available under `root`, as a view decoded from the flat arrays.
#pragma once
#include <vector>

#include "OctreeNode.h"
#include "rt_api.h"

class Octree {
public:
    Octree() : root(new OctreeNode()) {}
    ~Octree();
    Octree(const Octree&) = delete;
    Octree& operator=(const Octree&) = delete;

    // Octree.h:10: a triangle soup, three consecutive vertices per triangle; all of them end up in the tree.
    void init(Vertex* allVertices, int verticesCount);
    // The form Model uses: the model's own vertex and index arrays (and chunk boxes), so that the triangle ids in the flat
    // records are positions in the model's index list and the result can be passed back in rt_model.octree.  The last
    // triangle of the index list is left out, as every loop of the reference leaves it out (Model.cpp:282).
    void init(const rt_model& model);
    // Octree.h:11.  currentNode must be `root`: the flat arrays are built for the whole tree at once.
    void build(OctreeNode* currentNode);

    OctreeNode* root;
    int maxDepth = 0;            // 0 = the library's defaults (8 levels, 2 triangles per leaf)
    int leafTriangles = 0;
    bool pointerView = true;     // build() also expands the OctreeNode view under root (costs a copy of the vertices)

    const rt_flat_octree& flat() const { return flat_; }     // what the kernels read; empty before build()
    bool built() const { return built_; }
    void expandPointerView();    // decodes the flat arrays into OctreeNodes under root (no-op if done already)

    int nodeCount() const { return count(root, false); }     // nodes of the pointer view, leaves included
    int leafCount() const { return count(root, true); }      // non-empty leaves of the pointer view

private:
    void expand(OctreeNode* node, long long flatIndex);
    static int count(const OctreeNode* n, bool leavesOnly);

    rt_flat_octree flat_{};
    bool built_ = false, expanded_ = false;
    rt_model source_{};                  // the arrays the tree is built over
    std::vector<int> soupIndices_;       // init(vertices, count): 0, 1, 2, ... plus a throw-away last triangle
    std::vector<Vertex> leafVertices_;   // pointer view: vertices of all triangles in leaf order
};
