#include "Octree.h"

#include <algorithm>
#include <limits>
#include <vector>

namespace {
struct Centre { float c[3]; };

Centre triangleCentre(const Vertex* v) {
    Centre out;
    for (int a = 0; a < 3; a++) {
        float lo = std::min(v[0].position[a], std::min(v[1].position[a], v[2].position[a]));
        float hi = std::max(v[0].position[a], std::max(v[1].position[a], v[2].position[a]));
        out.c[a] = 0.5f * lo + 0.5f * hi;
    }
    return out;
}
}  // namespace

void Octree::subdivide(OctreeNode* node, int depth) {
    const int triangles = node->verticesCount / 3;
    node->isLeaf = true;
    if (triangles <= leafTriangles || depth >= maxDepth) return;

    std::vector<Centre> centres((size_t)triangles);
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    for (int t = 0; t < triangles; t++) {
        centres[(size_t)t] = triangleCentre(node->includedVertices + 3 * t);
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], centres[(size_t)t].c[a]); hi[a] = std::max(hi[a], centres[(size_t)t].c[a]); }
    }
    float mid[3];
    for (int a = 0; a < 3; a++) mid[a] = 0.5f * lo[a] + 0.5f * hi[a];
    auto octant = [&](const Centre& c) { return (c.c[0] > mid[0] ? 1 : 0) | (c.c[1] > mid[1] ? 2 : 0) | (c.c[2] > mid[2] ? 4 : 0); };

    int counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Centre& c : centres) counts[octant(c)]++;
    if (*std::max_element(counts, counts + 8) == triangles) return;   // cannot be separated

    node->isLeaf = false;
    node->children = new OctreeNode[8];
    int filled[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    Vertex* lists[8];
    for (int k = 0; k < 8; k++) lists[k] = counts[k] ? new Vertex[(size_t)counts[k] * 3] : nullptr;
    for (int t = 0; t < triangles; t++) {
        int k = octant(centres[(size_t)t]);
        for (int j = 0; j < 3; j++) lists[k][filled[k]++] = node->includedVertices[3 * t + j];
    }
    for (int k = 0; k < 8; k++) {
        node->children[k].add(lists[k], counts[k] * 3);
        node->children[k].ownsVertices = true;
        subdivide(&node->children[k], depth + 1);
    }
}

int Octree::count(const OctreeNode* n, bool leavesOnly) {
    if (!n) return 0;
    if (n->isLeaf) return (n->verticesCount > 0 || !leavesOnly) ? 1 : 0;
    int total = leavesOnly ? 0 : 1;
    for (int k = 0; k < 8; k++) total += count(&n->children[k], leavesOnly);
    return total;
}
