#include "Octree.h"

#include <cstring>
#include <iostream>

Octree::~Octree() {
    rt_free_octree(&flat_);
    delete root;
}

void Octree::init(Vertex* allVertices, int verticesCount) {
    // Octree.cpp:7-9 hands the vertices to the root, which grows its box over them (OctreeNode.cpp:8-14)
    root->add(allVertices, verticesCount);
    const int triangles = verticesCount / 3;
    soupIndices_.resize((size_t)triangles * 3 + 3);
    for (int i = 0; i < triangles * 3; i++) soupIndices_[(size_t)i] = i;
    // the renderer's loops never reach the last triangle of an index list: give them one to skip
    for (int k = 0; k < 3; k++) soupIndices_[(size_t)triangles * 3 + (size_t)k] = 0;
    std::memset(&source_, 0, sizeof source_);
    source_.vertices = reinterpret_cast<const rt_vertex*>(allVertices);
    source_.vertCount = verticesCount;
    source_.indices = soupIndices_.data();
    source_.indicesCount = triangles > 0 ? triangles * 3 + 3 : 0;
    built_ = expanded_ = false;
}

void Octree::init(const rt_model& model) {
    source_ = model;
    source_.octree = nullptr;
    for (int i = 0; i + 3 < model.indicesCount; i++) {
        const rt_vertex& v = model.vertices[model.indices[i]];
        root->boundingBox.extendBy(vec3(v.position.x, v.position.y, v.position.z));
    }
    built_ = expanded_ = false;
}

void Octree::build(OctreeNode* currentNode) {
    if (currentNode != root) { std::cerr << "Octree::build: only the whole tree can be built (pass root)" << std::endl; return; }
    rt_free_octree(&flat_);
    built_ = expanded_ = false;
    if (rt_build_octree(&source_, maxDepth, leafTriangles, &flat_) != RT_OK) {
        std::cerr << "Octree::build: " << rt_last_error() << std::endl;      // the reference's policy: report and carry on
        return;
    }
    built_ = true;
    if (pointerView) expandPointerView();
}

namespace {
struct FlatNode { uint32_t w[32]; };
inline float bitsToFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
}  // namespace

void Octree::expandPointerView() {
    if (!built_ || expanded_) return;
    expanded_ = true;
    delete[] root->children; root->children = nullptr;
    root->isLeaf = true;
    if (flat_.nodeCount == 0) return;
    // leaf-ordered copy of the triangles' vertices, which the leaves point into
    leafVertices_.resize((size_t)flat_.triangleRecords * 3);
    const uint32_t* tri = static_cast<const uint32_t*>(flat_.triangles);
    for (long long r = 0; r < flat_.triangleRecords; r++) {
        const uint32_t id = tri[12 * r + 3] & ((1u << 27) - 1u);
        for (int k = 0; k < 3; k++) {
            const rt_vertex& v = source_.vertices[source_.indices[3 * (size_t)id + (size_t)k]];
            std::memcpy(static_cast<void*>(&leafVertices_[(size_t)r * 3 + (size_t)k]), &v, sizeof(Vertex));
        }
    }
    expand(root, 0);
}

// Children of flat node `flatIndex` become the eight contiguous children of `node` (OctreeNode.h:13: children[8]): an
// internal slot recurses, a leaf slot points at its run of triangles, an unused slot stays an empty leaf.  Boxes are the
// decoded (outward-rounded) child boxes the traversal tests.
void Octree::expand(OctreeNode* node, long long flatIndex) {
    const FlatNode& n = static_cast<const FlatNode*>(flat_.nodes)[flatIndex];
    const uint32_t imask = n.w[5] & 0xffu, lmask = (n.w[5] >> 8) & 0xffu;
    node->isLeaf = false;
    node->children = new OctreeNode[8];
    const uint32_t* tri = static_cast<const uint32_t*>(flat_.triangles);
    uint32_t nextChild = n.w[4];
    for (int c = 0; c < 8; c++) {
        OctreeNode& ch = node->children[c];
        if (!(((imask | lmask) >> c) & 1u)) continue;
        for (int a = 0; a < 3; a++) {
            const float origin = bitsToFloat(n.w[a]);
            const float step = bitsToFloat(((n.w[3] >> (8 * a)) & 0xffu) << 23);
            const uint32_t qlo = (n.w[8 + 4 * a + c / 4] >> (8 * (c % 4))) & 0xffu, qhi = (n.w[8 + 4 * a + 2 + c / 4] >> (8 * (c % 4))) & 0xffu;
            ch.boundingBox.min[a] = origin + (float)qlo * step;
            ch.boundingBox.max[a] = origin + (float)qhi * step;
        }
        if ((imask >> c) & 1u) {
            expand(&ch, nextChild++);
            for (int k = 0; k < 8; k++) ch.verticesCount += ch.children[k].verticesCount;
        } else {
            const uint32_t first = n.w[20 + c];
            uint32_t last = first;
            while (!(tri[12 * (size_t)last + 3] & 0x80000000u)) last++;
            ch.includedVertices = &leafVertices_[(size_t)first * 3];
            ch.verticesCount = (int)(last - first + 1u) * 3;
        }
    }
}

int Octree::count(const OctreeNode* n, bool leavesOnly) {
    if (!n) return 0;
    if (n->isLeaf) return (n->verticesCount > 0 || !leavesOnly) ? 1 : 0;
    int total = leavesOnly ? 0 : 1;
    for (int k = 0; k < 8; k++) total += count(&n->children[k], leavesOnly);
    return total;
}
