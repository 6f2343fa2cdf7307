// Host mirror of OctreeNode (OctreeNode.h:6-20, OctreeNode.cpp:3-14): bounding box, leaf flag, eight contiguous
// children, and the vertices (three per triangle) the node holds.
#pragma once
#include "AABB.h"
#include "utils.h"

class OctreeNode {
public:
    OctreeNode() {}
    ~OctreeNode() { delete[] children; if (ownsVertices) delete[] includedVertices; }
    OctreeNode(const OctreeNode&) = delete;
    OctreeNode& operator=(const OctreeNode&) = delete;

    // OctreeNode.cpp:8-14: takes the vertex list and grows the box over it
    void add(Vertex* vertices, int count) {
        includedVertices = vertices;
        verticesCount = count;
        for (int i = 0; i < count; i++) boundingBox.extendBy(vertices[i].position);
    }

    AABB boundingBox;
    bool isLeaf = true;
    OctreeNode* children = nullptr;
    int verticesCount = 0;
    Vertex* includedVertices = nullptr;
    bool ownsVertices = false;
};
