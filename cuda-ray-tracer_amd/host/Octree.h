// Host mirror of Octree (Octree.h:6-17).  init() takes a triangle soup (three consecutive vertices per triangle);
// build() subdivides SPATIALLY: the reference's as-written build (Octree.cpp:11-43) deals the vertex list out
// sequentially and is never called, so only its interface is kept.  The split rule is the one the renderer's device
// octree uses (octant of the triangle-centre relative to the middle of the centre bounds; leaf at <= leafTriangles
// triangles or maxDepth).  The renderer itself builds its flat device octree inside rt_upload_scene; this
// pointer tree exists for host code that wants to walk `root`.
#pragma once
#include "OctreeNode.h"

class Octree {
public:
    Octree() : root(new OctreeNode()) {}
    ~Octree() { delete root; }
    Octree(const Octree&) = delete;
    Octree& operator=(const Octree&) = delete;

    void init(Vertex* allVertices, int verticesCount) { root->add(allVertices, verticesCount); }
    void build(OctreeNode* currentNode) { subdivide(currentNode, 0); }

    OctreeNode* root;
    int maxDepth = 8;
    int leafTriangles = 4;

    int nodeCount() const { return count(root, false); }
    int leafCount() const { return count(root, true); }

private:
    void subdivide(OctreeNode* node, int depth);
    static int count(const OctreeNode* n, bool leavesOnly);
};
