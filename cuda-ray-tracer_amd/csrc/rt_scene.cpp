// Scene flattening and octree construction (host).
//
// Replaces the reference's pointer-linked managed-memory octree (Octree.cpp:11-43, OctreeNode.cpp:3-14), which is
// never built nor traversed there, by a spatial octree that is emitted directly as the flat 128-byte node array
// the kernels read.  The octree is only an accelerator: a mesh hit is defined by the reference's whole-model box
// gate plus its linear triangle loop (Model.cpp:275-317), and the traversal kernel reproduces exactly that result.
#include "rt_scene.h"

#include "rt_node_codec.h"

#include <algorithm>
#include <array>
#include <thread>
#include <memory>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rt {

namespace {


inline float bitsFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline F4 f4(float x, float y, float z, float w) { return F4{x, y, z, w}; }
inline F4 f4(const rt_vec3& v, float w) { return F4{v.x, v.y, v.z, w}; }
inline bool finite3(const rt_vec3& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }

struct Box {
    float lo[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
    float hi[3] = {-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity()};
    void grow(const float p[3]) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
};

// Triangle as the builder sees it.
struct BuildTri {
    Box box;
    float centre[3];
};

// Temporary pointer tree; flattened breadth-first afterwards so that siblings are contiguous.
struct TreeNode {
    Box box;
    int child[8];
    int first = 0, count = 0;   // leaf: slice of the permutation array
    bool leaf = true;
};

class OctreeBuilder {
public:
    OctreeBuilder(const std::vector<BuildTri>& tris, const OctreeParams& p) : tris_(tris), params_(p) {
        perm_.resize(tris.size());
        for (size_t i = 0; i < tris.size(); i++) perm_[i] = (int)i;
        scratch_.resize(tris.size());
    }

    int build() {
        if (tris_.empty()) return -1;
        int root;
        if (params_.sah) {
            // binary tree by surface-area heuristic; widen() then folds it into nodes of up to eight children
            root = buildSah(0, (int)tris_.size(), 0);
            if (params_.optimalCollapse) prepareCollapse(root);
            widen(root);
            deepest = 0;
            capDepth(root, 0);
        } else {
            root = buildRange(0, (int)tris_.size(), 0);
            if (params_.widen) widen(root);
        }
        return root;
    }

    std::vector<TreeNode> nodes;
    std::vector<int> perm_;
    int deepest = 0;

private:
    int buildRange(int first, int count, int depth) {
        int self = (int)nodes.size();
        nodes.emplace_back();
        {
            TreeNode& n = nodes[self];
            for (int c = 0; c < 8; c++) n.child[c] = -1;
            n.first = first; n.count = count;
            for (int i = 0; i < count; i++) n.box.grow(tris_[perm_[first + i]].box);
        }
        deepest = std::max(deepest, depth);
        if (count <= params_.leafTriangles || depth >= params_.maxDepth) return self;

        // Split point: middle of the triangles' centre bounds.
        Box cb;
        for (int i = 0; i < count; i++) cb.grow(tris_[perm_[first + i]].centre);
        float mid[3];
        for (int a = 0; a < 3; a++) mid[a] = 0.5f * cb.lo[a] + 0.5f * cb.hi[a];

        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < count; i++) cnt[octant(tris_[perm_[first + i]].centre, mid)]++;
        int biggest = *std::max_element(cnt, cnt + 8);
        if (biggest == count) return self;   // all centres on one side of every plane: cannot be split further

        int off[8], pos[8];
        off[0] = 0;
        for (int c = 1; c < 8; c++) off[c] = off[c - 1] + cnt[c - 1];
        std::memcpy(pos, off, sizeof pos);
        for (int i = 0; i < count; i++) {
            int t = perm_[first + i];
            scratch_[first + pos[octant(tris_[t].centre, mid)]++] = t;
        }
        std::copy(scratch_.begin() + first, scratch_.begin() + first + count, perm_.begin() + first);

        nodes[self].leaf = false;
        for (int c = 0; c < 8; c++) {
            if (cnt[c] == 0) continue;
            int childIdx = buildRange(first + off[c], cnt[c], depth + 1);
            nodes[self].child[c] = childIdx;
        }
        return self;
    }

    // Binned surface-area split of the range, recursively: 16 bins per axis over the bounds of the triangle centres, the
    // plane with the least area(left) x count(left) + area(right) x count(right) wins.  Ranges whose centres coincide,
    // and ranges nested deeper than 64 splits, are halved by index.
    int buildSah(int first, int count, int depth) {
        const int self = (int)nodes.size();
        nodes.emplace_back();
        {
            TreeNode& n = nodes[self];
            for (int c = 0; c < 8; c++) n.child[c] = -1;
            n.first = first; n.count = count;
            for (int i = 0; i < count; i++) n.box.grow(tris_[perm_[first + i]].box);
        }
        if (count <= params_.sahLeafTriangles || count < 2) return self;
        Box cb;
        for (int i = 0; i < count; i++) cb.grow(tris_[perm_[first + i]].centre);
        constexpr int kBins = 16;
        int bestAxis = -1, bestBin = 0;
        float bestCost = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3 && depth < 64; a++) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Box binBox[kBins];
            int binCount[kBins] = {};
            const float scale = (float)kBins / ext;
            for (int i = 0; i < count; i++) {
                const BuildTri& t = tris_[perm_[first + i]];
                int b = (int)((t.centre[a] - cb.lo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                binCount[b]++;
                binBox[b].grow(t.box);
            }
            float leftArea[kBins];
            int leftCount[kBins];
            Box acc;
            int n = 0;
            for (int b = 0; b < kBins; b++) {
                if (binCount[b]) acc.grow(binBox[b]);
                n += binCount[b];
                leftArea[b] = n ? area(acc) : 0.0f; leftCount[b] = n;
            }
            acc = Box(); n = 0;
            for (int b = kBins - 1; b >= 1; b--) {
                if (binCount[b]) acc.grow(binBox[b]);
                n += binCount[b];
                if (leftCount[b - 1] == 0 || n == 0) continue;
                const float cost = leftArea[b - 1] * (float)leftCount[b - 1] + area(acc) * (float)n;
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b; }     // left = bins below bestBin
            }
        }
        int mid = count / 2;
        if (bestAxis >= 0) {
            const float lo = cb.lo[bestAxis], scale = (float)kBins / (cb.hi[bestAxis] - cb.lo[bestAxis]);
            auto it = std::partition(perm_.begin() + first, perm_.begin() + first + count, [&](int t) {
                int b = (int)((tris_[(size_t)t].centre[bestAxis] - lo) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b < bestBin;
            });
            const int left = (int)(it - (perm_.begin() + first));
            if (left > 0 && left < count) mid = left;
        }
        nodes[self].leaf = false;
        const int l = buildSah(first, mid, depth + 1);
        const int r = buildSah(first + mid, count - mid, depth + 1);
        nodes[self].child[0] = l; nodes[self].child[1] = r;
        return self;
    }

    // Nesting deeper than the caller's cap: the subtree becomes one leaf (its triangles are one slice of the permutation).
    void capDepth(int ni, int depth) {
        deepest = std::max(deepest, depth);
        if (nodes[ni].leaf) return;
        if (depth >= params_.maxDepth) { nodes[ni].leaf = true; for (int c = 0; c < 8; c++) nodes[ni].child[c] = -1; return; }
        for (int c = 0; c < 8; c++) if (nodes[ni].child[c] >= 0) capDepth(nodes[ni].child[c], depth + 1);
    }

    static float area(const Box& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    // A node of the octree of a surface has about half of its eight slots in use.  This pass fills the free ones: as
    // long as there is room, the internal child with the largest box (the one a ray is most likely to enter) is replaced
    // by its own children.  The tree gets shallower and a ray visits fewer nodes for the same triangles.  Afterwards the
    // children are dealt to the slots so that slot index bit a set still means "towards +a within the node", which is
    // what the traversal's front-to-back order (slot ^ ray octant) relies on; the order is a heuristic, hits do not
    // depend on it.
    void widen(int ni) {
        if (nodes[ni].leaf) return;
        std::vector<int> kids;
        if (!collapseCost_.empty() && nodes[ni].child[0] >= 0 && nodes[ni].child[1] >= 0) {
            float cost;
            const int k = bestSplit(ni, 8, cost);
            collectKids(nodes[ni].child[0], k, kids);
            collectKids(nodes[ni].child[1], 8 - k, kids);
        } else {
            for (int c = 0; c < 8; c++) if (nodes[ni].child[c] >= 0) kids.push_back(nodes[ni].child[c]);
        }
        for (; collapseCost_.empty();) {
            int pick = -1;
            float pickArea = -1.0f;
            for (size_t i = 0; i < kids.size(); i++) {
                const TreeNode& k = nodes[kids[i]];
                if (k.leaf) continue;
                int grand = 0;
                for (int c = 0; c < 8; c++) grand += k.child[c] >= 0 ? 1 : 0;
                if ((int)kids.size() - 1 + grand > 8) continue;
                const float a = area(k.box);
                if (a > pickArea) { pickArea = a; pick = (int)i; }
            }
            if (pick < 0) break;
            const int gone = kids[(size_t)pick];
            kids.erase(kids.begin() + pick);
            for (int c = 0; c < 8; c++) if (nodes[gone].child[c] >= 0) kids.push_back(nodes[gone].child[c]);
        }
        // slots: best total of (child centre - node centre) . (slot direction), greedily
        const Box& nb = nodes[ni].box;
        struct Pair { float score; int kid, slot; };
        std::vector<Pair> pairs;
        for (size_t i = 0; i < kids.size(); i++) {
            const Box& kb = nodes[kids[i]].box;
            float rel[3];
            for (int a = 0; a < 3; a++) {
                const float ext = nb.hi[a] - nb.lo[a];
                rel[a] = ext > 0 ? ((0.5f * kb.lo[a] + 0.5f * kb.hi[a]) - (0.5f * nb.lo[a] + 0.5f * nb.hi[a])) / ext : 0.0f;
            }
            for (int s = 0; s < 8; s++) {
                const float sc = ((s & 1) ? rel[0] : -rel[0]) + ((s & 2) ? rel[1] : -rel[1]) + ((s & 4) ? rel[2] : -rel[2]);
                pairs.push_back(Pair{sc, (int)i, s});
            }
        }
        std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.score > b.score; });
        int slotOf[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        bool kidDone[8] = {false, false, false, false, false, false, false, false};
        for (const Pair& p : pairs) {
            if (kidDone[p.kid] || slotOf[p.slot] >= 0) continue;
            slotOf[p.slot] = kids[(size_t)p.kid];
            kidDone[p.kid] = true;
        }
        for (int c = 0; c < 8; c++) nodes[ni].child[c] = slotOf[c];
        for (int c = 0; c < 8; c++) if (slotOf[c] >= 0) widen(slotOf[c]);
    }

    // Which binary nodes stay nodes: every node that stays costs its surface area (the chance that a ray visits it), a node
    // holds at most eight children, leaves cost the same wherever they hang.  collapseCost_[n][j - 1] = least total area below
    // binary node n when n may put up to j children into its parent: either n stays (one child: area(n) + the best use of its
    // own eight places) or n is opened and its two children share the j places.
    void prepareCollapse(int root) {
        collapseCost_.assign(nodes.size(), std::array<float, 8>());
        // children have larger indices than their parents (buildSah numbers a node before it descends)
        for (int n = (int)nodes.size() - 1; n >= root; n--) {
            std::array<float, 8>& cost = collapseCost_[(size_t)n];
            if (nodes[(size_t)n].leaf) { cost.fill(0.0f); continue; }
            float open8;
            bestSplit(n, 8, open8);
            const float stay = area(nodes[(size_t)n].box) + open8;
            cost[0] = stay;
            for (int j = 2; j <= 8; j++) {
                float open;
                bestSplit(n, j, open);
                cost[(size_t)j - 1] = std::min(open, stay);
            }
        }
    }

    // Places for the left child when internal binary node n is opened into j >= 2 places; `cost` = total area below.
    int bestSplit(int n, int j, float& cost) const {
        const std::array<float, 8>& l = collapseCost_[(size_t)nodes[(size_t)n].child[0]];
        const std::array<float, 8>& r = collapseCost_[(size_t)nodes[(size_t)n].child[1]];
        cost = std::numeric_limits<float>::infinity();
        int bestK = 1;
        for (int k = 1; k < j; k++) {
            const float c = l[(size_t)k - 1] + r[(size_t)(j - k) - 1];
            if (c < cost) { cost = c; bestK = k; }
        }
        return bestK;
    }

    // Binary node n with up to j places in the wide node being assembled: n itself, or the kids of its opened children.
    void collectKids(int n, int j, std::vector<int>& kids) const {
        if (nodes[(size_t)n].leaf || j < 2) { kids.push_back(n); return; }
        float open, open8;
        const int k = bestSplit(n, j, open);
        bestSplit(n, 8, open8);
        if (!(open < area(nodes[(size_t)n].box) + open8)) { kids.push_back(n); return; }
        collectKids(nodes[(size_t)n].child[0], k, kids);
        collectKids(nodes[(size_t)n].child[1], j - k, kids);
    }

    static int octant(const float c[3], const float mid[3]) {
        return (c[0] > mid[0] ? 1 : 0) | (c[1] > mid[1] ? 2 : 0) | (c[2] > mid[2] ? 4 : 0);
    }

    const std::vector<BuildTri>& tris_;
    OctreeParams params_;
    std::vector<int> scratch_;
    std::vector<std::array<float, 8>> collapseCost_;
};

}  // namespace

int64_t reachableTriangles(int64_t indicesCount) {
    if (indicesCount <= 3) return 0;
    return (indicesCount - 3 + 2) / 3;
}

void packMaterial(const rt_material& m, int objectType, uint32_t flags, F4 out[3]) {
    out[0] = f4(m.ambientColour, m.ambient);
    out[1] = f4(m.diffuseColour, m.diffuse);
    out[2] = f4(m.specular, m.shininess, bitsFloat((uint32_t)objectType), bitsFloat(flags));
}

static bool isIgnoredPosition(const rt_vec3& p) { return p.x == -999.0f && p.y == -999.0f && p.z == -999.0f; }

// Validates a model's arrays and returns the bounds of its reachable triangles and the padding of its child boxes.
// Child boxes are padded so that the approximate slab test of the traversal can never reject a box whose
// triangle the exact intersection routine would accept.  That test forms (plane - ray origin) * (1 / d) per plane: its
// error on the plane's axis is below 3e-7 of the distance between the ray origin and the node (five roundings of 2^-24 to
// 2^-23 each), whatever the absolute coordinates are.  A hit that counts lies within 999 units of the ray origin, a node
// spans at most the model, so the error stays below 3e-7 * (999 + 2 maxAbs); the pad is at least twice that.
// (The mesh mode without a t > 0 test also accepts hits behind the ray origin at any distance: rt_api.cpp limits how far
// the camera may be from the meshes in that mode.)
static bool modelBounds(const rt_model& m, int modelIndex, Box& tight, float& pad, std::string& err) {
    const int64_t nTri = reachableTriangles(m.indicesCount);
    if (nTri > 0 && (!m.vertices || !m.indices)) { err = "model " + std::to_string(modelIndex) + ": null vertex or index array"; return false; }
    if (nTri > (int64_t)kTriIdMask) { err = "model " + std::to_string(modelIndex) + ": too many triangles"; return false; }
    float maxAbs = 0.0f;
    for (int64_t i = 0; i < 3 * nTri; i++) {
        const int32_t ix = m.indices[i];
        if (ix < 0 || ix >= m.vertCount) { err = "model " + std::to_string(modelIndex) + ": index out of range"; return false; }
        const rt_vec3& p = m.vertices[ix].position;
        if (!finite3(p)) { err = "model " + std::to_string(modelIndex) + ": non-finite vertex position"; return false; }
        const float pp[3] = {p.x, p.y, p.z};
        tight.grow(pp);
        maxAbs = std::max(maxAbs, std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
    }
    pad = std::max(1e-3f, 4e-6f * maxAbs);
    return true;
}

// Bounds and centre of every reachable triangle of a model: what the tree builder works on.
static void fillBuildTris(const rt_model& m, int64_t nTri, std::vector<BuildTri>& btris) {
    btris.assign((size_t)nTri, BuildTri());
    for (int64_t t = 0; t < nTri; t++) {
        const int32_t* ix = m.indices + 3 * t;
        BuildTri& bt = btris[(size_t)t];
        for (int k = 0; k < 3; k++) {
            const rt_vec3& v = m.vertices[ix[k]].position;
            const float p[3] = {v.x, v.y, v.z};
            bt.box.grow(p);
        }
        for (int ax = 0; ax < 3; ax++) bt.centre[ax] = 0.5f * bt.box.lo[ax] + 0.5f * bt.box.hi[ax];
    }
}

// The tree of one model, built ahead of flattenModel: the models of a scene are independent, so their trees are built on
// several threads (144 models with 10.3 M triangles: 16 threads instead of one), and emitted one after the other.
struct HostTree {
    std::vector<BuildTri> tris;
    std::unique_ptr<OctreeBuilder> builder;
    int root = -1;
};

static bool buildTreesAhead(const rt_scene_desc& s, const OctreeParams& params, std::vector<HostTree>& trees) {
    std::vector<std::pair<int64_t, int>> work;      // (triangles, model), largest first
    int64_t total = 0;
    for (int i = 0; i < s.modelCount; i++) {
        const rt_model& m = s.models[i];
        const int64_t nTri = reachableTriangles(m.indicesCount);
        if (nTri == 0 || (m.octree && m.octree->nodeCount > 0)) continue;
        work.emplace_back(nTri, i);
        total += nTri;
    }
    int threads = (int)std::min<size_t>({work.size(), (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16});
    if (const char* e = std::getenv("RT_BUILD_THREADS")) threads = std::max(1, std::min(threads, std::atoi(e)));
    if (threads < 2 || total < 50000) return true;      // flattenModel builds the trees itself
    std::sort(work.begin(), work.end(), [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto run = [&]() {
        for (size_t k = next.fetch_add(1); k < work.size() && !failed.load(); k = next.fetch_add(1)) {
            try {
                HostTree& t = trees[(size_t)work[k].second];
                fillBuildTris(s.models[work[k].second], work[k].first, t.tris);
                t.builder.reset(new OctreeBuilder(t.tris, params));
                t.root = t.builder->build();
            } catch (...) {
                failed.store(true);
            }
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int k = 1; k < threads; k++) pool.emplace_back(run);
    } catch (...) {
        // fewer threads than asked for: the ones that started and this one share the work
    }
    run();
    for (std::thread& t : pool) t.join();
    return !failed.load();
}

// Emits the octree and triangle records of one model into the shared arrays.  `rootIndex` is the node the forest
// reserved for the model's root (unused for a model without reachable triangles); `tight` / `pad` come from modelBounds.
static bool flattenModel(const rt_model& m, int modelIndex, const OctreeParams& params, uint32_t rootIndex, const Box& tight,
                         float pad, bool onDevice, HostTree& ahead, FlatScene& out, std::string& err) {
    const int64_t nTri = reachableTriangles(m.indicesCount);
    const uint32_t triBase = (uint32_t)out.triangleCount;

    (void)err;
    // An octree the caller built beforehand with rt_build_octree for this very model (the host's Octree class) is spliced in
    // below instead of building one here.
    const rt_flat_octree* pre = (!onDevice && m.octree && m.octree->nodeCount > 0) ? m.octree : nullptr;
    if (pre) {
        if (!pre->nodes || !pre->triangles || pre->triangleRecords != nTri || pre->nodeCount > (int64_t)1 << 30 || pre->depth < 0 || pre->depth > kMaxModelOctreeDepth) {
            err = "model " + std::to_string(modelIndex) + ": the supplied octree does not belong to this model";
            return false;
        }
    }
    // the tree may have been built ahead (buildTreesAhead); otherwise it is built here
    std::vector<BuildTri> btris;
    if (!onDevice && !pre && !ahead.builder) fillBuildTris(m, nTri, btris);
    for (int64_t t = 0; t < nTri && !onDevice; t++) {
        const int32_t* ix = m.indices + 3 * t;
        const rt_vertex& a = m.vertices[ix[0]];
        const rt_vertex& b = m.vertices[ix[1]];
        const rt_vertex& c = m.vertices[ix[2]];

        // shading attributes, original triangle order
        F4 rec[kAttrFloat4s];
        rec[0] = f4(a.normal, a.ambient.x);
        rec[1] = f4(b.normal, a.ambient.y);
        rec[2] = f4(c.normal, a.ambient.z);
        rec[3] = f4(b.ambient, a.diffuse.x);
        rec[4] = f4(c.ambient, a.diffuse.y);
        rec[5] = f4(b.diffuse, a.diffuse.z);
        rec[6] = f4(c.diffuse, 0.0f);
        out.attrs.insert(out.attrs.end(), rec, rec + kAttrFloat4s);
    }

    // Chunk membership for the eight-box mode (Model.cpp:62-84): triangle t sits in chunk t / perBox at slot t % perBox, and
    // the reference's loop over a chunk stops one slot short of its end.  This only holds if the caller's chunk index
    // lists are the sequential deal of `indices` that the reference constructor produces; remember if they are not.
    const int64_t perBoxTris = (int64_t)std::ceil((double)(m.indicesCount / 3) / 8.0);
    std::vector<uint32_t> chunkTag((size_t)nTri, 0u);
    for (int64_t t = 0; t < nTri; t++) {
        const int64_t b = perBoxTris > 0 ? t / perBoxTris : 0, slot = perBoxTris > 0 ? t % perBoxTris : 0;
        bool ok = b < 8 && m.boundingBoxes[b].includedModelIndices && (int64_t)m.boundingBoxes[b].indicesCount >= 3 * (slot + 1);
        if (ok) {
            const int32_t* ci = m.boundingBoxes[b].includedModelIndices + 3 * slot;
            const int32_t* ix = m.indices + 3 * t;
            ok = ci[0] == ix[0] && ci[1] == ix[1] && ci[2] == ix[2];
        }
        if (!ok) { out.chunksSequential = false; break; }
        const bool lastSlot = 3 * (slot + 1) >= (int64_t)m.boundingBoxes[b].indicesCount;   // for (j = 0; j < indicesCount - 3; j += 3) never reaches it
        chunkTag[(size_t)t] = ((uint32_t)b << kTriChunkShift) | (lastSlot ? kTriChunkDropped : 0u);
    }
    for (int b = 0; b < 8; b++) {
        out.mdlChunkMin.push_back(f4(m.boundingBoxes[b].min, 0.0f));
        out.mdlChunkMax.push_back(f4(m.boundingBoxes[b].max, 0.0f));
    }

    if (onDevice && nTri > 0) {
        // octree, triangle and attribute records of this model are produced on the GPU (rt_build.hip)
        DeviceBuildModel dm;
        dm.model = modelIndex; dm.rootIndex = rootIndex; dm.triStart = triBase; dm.triCount = (uint32_t)nTri;
        for (int a = 0; a < 3; a++) { dm.lo[a] = tight.lo[a]; dm.hi[a] = tight.hi[a]; }
        dm.pad = pad;
        dm.perBoxTris = (int32_t)perBoxTris;
        for (int b = 0; b < 8; b++) dm.chunkIndices[b] = m.boundingBoxes[b].indicesCount;
        out.deviceModels.push_back(dm);
    }

    OctreeBuilder ownBuilder(btris, params);
    OctreeBuilder& builder = ahead.builder ? *ahead.builder : ownBuilder;
    int root = (onDevice || pre) ? -1 : (ahead.builder ? ahead.root : builder.build());

    std::vector<int> leafIds;      // reused by every leaf of the model
    auto emitLeaf = [&](const TreeNode& n) -> uint32_t {
        uint32_t start = (uint32_t)(out.tris.size() / 3);
        leafIds.assign(builder.perm_.begin() + n.first, builder.perm_.begin() + n.first + n.count);
        std::vector<int>& ids = leafIds;
        std::sort(ids.begin(), ids.end());
        for (size_t i = 0; i < ids.size(); i++) {
            const int32_t* ix = m.indices + 3 * (int64_t)ids[i];
            const rt_vec3& p0 = m.vertices[ix[0]].position;
            const rt_vec3& p1 = m.vertices[ix[1]].position;
            const rt_vec3& p2 = m.vertices[ix[2]].position;
            uint32_t id = (uint32_t)ids[i] | chunkTag[(size_t)ids[i]] | (i + 1 == ids.size() ? kTriLast : 0u);
            out.tris.push_back(f4(p0.x, p0.y, p0.z, bitsFloat(id)));
            out.tris.push_back(f4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, bitsFloat((uint32_t)modelIndex)));   // Model.cpp:137
            out.tris.push_back(f4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, bitsFloat(triBase)));   // Model.cpp:138
        }
        out.leafCount++;
        return start;
    };

    uint32_t rootRef = kNoRoot;
    if (root >= 0) {
        const std::vector<TreeNode>& T = builder.nodes;
        const float inf = std::numeric_limits<float>::infinity();
        rootRef = rootIndex;
        if (T[root].leaf) {
            // a model small enough for one leaf still gets a root node, with that leaf as its only child
            float lo[3][8], hi[3][8];
            uint32_t leafRef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = 0; c < 8; c++) for (int a = 0; a < 3; a++) { lo[a][c] = inf; hi[a][c] = -inf; }
            for (int a = 0; a < 3; a++) { lo[a][0] = tight.lo[a] - pad; hi[a][0] = tight.hi[a] + pad; }
            leafRef[0] = emitLeaf(T[root]);
            encodeNode(&out.nodes[(size_t)rootIndex * kNodeFloat4s], lo, hi, leafRef, 0u, 0u, 1u, (uint32_t)modelIndex);
        } else {
            // breadth-first flattening: (tree node, flat node index)
            std::vector<std::pair<int, uint32_t>> queue;
            queue.emplace_back(root, rootIndex);
            for (size_t q = 0; q < queue.size(); q++) {
                const TreeNode& n = T[queue[q].first];
                const uint32_t self = queue[q].second;
                float lo[3][8], hi[3][8];
                uint32_t leafRef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t imask = 0, lmask = 0;
                uint32_t childBase = (uint32_t)(out.nodes.size() / kNodeFloat4s);
                int internalChildren = 0;
                for (int c = 0; c < 8; c++) {
                    if (n.child[c] >= 0 && !T[n.child[c]].leaf) internalChildren++;
                }
                out.nodes.resize(out.nodes.size() + (size_t)internalChildren * kNodeFloat4s);
                uint32_t next = childBase;
                for (int c = 0; c < 8; c++) {
                    for (int a = 0; a < 3; a++) { lo[a][c] = inf; hi[a][c] = -inf; }
                    if (n.child[c] < 0) continue;
                    const TreeNode& ch = T[n.child[c]];
                    for (int a = 0; a < 3; a++) { lo[a][c] = ch.box.lo[a] - pad; hi[a][c] = ch.box.hi[a] + pad; }
                    if (ch.leaf) { lmask |= 1u << c; leafRef[c] = emitLeaf(ch); }
                    else { imask |= 1u << c; queue.emplace_back(n.child[c], next++); }
                }
                encodeNode(&out.nodes[(size_t)self * kNodeFloat4s], lo, hi, leafRef, childBase, imask, lmask,
                          q == 0 ? (uint32_t)modelIndex : kNoModel);
            }
        }
        out.maxDepth = std::max(out.maxDepth, std::max(1, builder.deepest));
    }

    if (pre) {
        // Records of a standalone build: node 0 is the root, the others follow breadth-first with absolute child
        // references; triangle records are numbered from 0 and tagged with model 0.  Here the root goes to the node the
        // forest reserved, the rest to the end of the array, and every reference moves along.
        const uint32_t base = (uint32_t)(out.nodes.size() / kNodeFloat4s);
        const uint32_t triOffset = (uint32_t)(out.tris.size() / 3);
        const uint32_t n = (uint32_t)pre->nodeCount;
        out.nodes.resize(out.nodes.size() + (size_t)(n - 1) * kNodeFloat4s);
        const uint32_t* src = static_cast<const uint32_t*>(pre->nodes);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t w[32];
            std::memcpy(w, src + 32 * (size_t)i, sizeof w);
            const uint32_t imask = w[5] & 0xffu, lmask = (w[5] >> 8) & 0xffu;
            if (imask) {
                if (w[4] < 1u || w[4] >= n) { err = "model " + std::to_string(modelIndex) + ": supplied octree has a child reference out of range"; return false; }
                w[4] = base + (w[4] - 1u);
            }
            w[6] = i == 0 ? (uint32_t)modelIndex : kNoModel;
            for (int c = 0; c < 8; c++) {
                if (!((lmask >> c) & 1u)) continue;
                if (w[20 + c] >= (uint32_t)nTri) { err = "model " + std::to_string(modelIndex) + ": supplied octree has a leaf reference out of range"; return false; }
                w[20 + c] += triOffset;
            }
            std::memcpy(&out.nodes[(size_t)(i == 0 ? rootIndex : base + i - 1u) * kNodeFloat4s], w, sizeof w);
        }
        const F4* tsrc = static_cast<const F4*>(pre->triangles);
        for (int64_t t = 0; t < nTri; t++) {
            F4 a = tsrc[3 * t], b = tsrc[3 * t + 1], c = tsrc[3 * t + 2];
            b.w = bitsFloat((uint32_t)modelIndex);
            c.w = bitsFloat(triBase);
            out.tris.push_back(a); out.tris.push_back(b); out.tris.push_back(c);
        }
        out.leafCount += pre->leafCount;
        out.maxDepth = std::max(out.maxDepth, std::max(1, (int)pre->depth));
        out.prebuiltOctrees++;
        rootRef = rootIndex;
    }

    F4 mat[3];
    packMaterial(m.mat, m.objectType, m.debug ? 1u : 0u, mat);
    out.mdlMat.insert(out.mdlMat.end(), mat, mat + 3);
    out.mdlGateMin.push_back(f4(m.bbMin, 0.0f));
    out.mdlGateMax.push_back(f4(m.bbMax, 0.0f));
    if (nTri > 0) {
        // When does a ray that meets the tight box also pass the reference's gate (AABB::hit on the model's own box, which holds
        // the world origin and is padded by 0.1 wherever a vertex extended it)?  If the gate box surrounds the tight box by a
        // margin m on every side, a ray found to meet the tight box within [0, tmax] has a point at a parameter >= 0 that lies
        // inside the gate box by nearly m on every axis, so each of the gate's three parameter intervals contains that
        // parameter with a slack of m / |d_a|, while the gate's six quotients carry an error below 2e-7 D / |d_a| (D = the
        // largest distance between the ray origin and a gate plane along an axis) - and the tight test itself less than 1e-6 D
        // in position.  With m >= 8e-6 D every comparison of AABB::hit is decided with room to spare and the gate accepts the
        // ray; D <= (largest gate coordinate) + (largest origin coordinate), so the kernel needs one comparison of the
        // latter with the limit below to know the gate's answer without evaluating it (finite reciprocal directions only).
        float margin = std::numeric_limits<float>::infinity(), gateAbs = 0.0f;
        const float bbLo[3] = {m.bbMin.x, m.bbMin.y, m.bbMin.z}, bbHi[3] = {m.bbMax.x, m.bbMax.y, m.bbMax.z};
        for (int a = 0; a < 3; a++) {
            margin = std::min(margin, std::min((tight.lo[a] - pad) - bbLo[a], bbHi[a] - (tight.hi[a] + pad)));
            gateAbs = std::max(gateAbs, std::max(std::fabs(bbLo[a]), std::fabs(bbHi[a])));
        }
        const float impliedLimit = (margin > 0.0f && std::isfinite(margin) && std::isfinite(gateAbs)) ? margin / 8e-6f - gateAbs
                                                                                                   : -std::numeric_limits<float>::infinity();
        out.mdlTightMin.push_back(f4(tight.lo[0] - pad, tight.lo[1] - pad, tight.lo[2] - pad, impliedLimit));
        out.mdlTightMax.push_back(f4(tight.hi[0] + pad, tight.hi[1] + pad, tight.hi[2] + pad, 0.0f));
    } else {
        const float inf = std::numeric_limits<float>::infinity();
        out.mdlTightMin.push_back(f4(inf, inf, inf, 0.0f));
        out.mdlTightMax.push_back(f4(-inf, -inf, -inf, 0.0f));
    }
    if (onDevice && nTri > 0) rootRef = rootIndex;
    out.mdlInfo.push_back(rootRef);
    out.mdlInfo.push_back(triBase);
    out.mdlInfo.push_back(0);
    out.mdlInfo.push_back(0);
    out.triangleCount += nTri;
    return true;
}

namespace {

struct ForestItem {
    int model;
    Box box;          // padded bounds of the model's reachable triangles
    float centre[3];
};

// Fills node `self` for items[first, first + count).  A single item means `self` is that model's root, which
// flattenModel writes later.  Otherwise the items are dealt into up to eight groups - by octant around the middle
// of their centres while that separates them and the forest is shallow, by equal index ranges after that - and every
// group becomes one internal child; the children of a node are contiguous.
void buildForest(std::vector<ForestItem>& items, int first, int count, uint32_t self, int depth, FlatScene& out,
                 std::vector<uint32_t>& rootIndex, int& deepest) {
    if (count == 1) { rootIndex[(size_t)items[(size_t)first].model] = self; return; }
    deepest = std::max(deepest, depth + 1);

    int slotOf[8], groupFirst[8], groupCount[8];
    int groups = 0;
    bool spatial = false;
    if (depth < 4) {
        Box cb;
        for (int i = 0; i < count; i++) cb.grow(items[(size_t)(first + i)].centre);
        float mid[3];
        for (int a = 0; a < 3; a++) mid[a] = 0.5f * cb.lo[a] + 0.5f * cb.hi[a];
        auto octant = [&](const ForestItem& it) {
            return (it.centre[0] > mid[0] ? 1 : 0) | (it.centre[1] > mid[1] ? 2 : 0) | (it.centre[2] > mid[2] ? 4 : 0);
        };
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < count; i++) cnt[octant(items[(size_t)(first + i)])]++;
        if (*std::max_element(cnt, cnt + 8) < count) {
            spatial = true;
            std::stable_sort(items.begin() + first, items.begin() + first + count,
                             [&](const ForestItem& a, const ForestItem& b) { return octant(a) < octant(b); });
            int at = first;
            for (int c = 0; c < 8; c++) {
                if (cnt[c] == 0) continue;
                slotOf[groups] = c; groupFirst[groups] = at; groupCount[groups] = cnt[c];
                at += cnt[c]; groups++;
            }
        }
    }
    if (!spatial) {
        groups = count < 8 ? count : 8;
        for (int g = 0; g < groups; g++) {
            const int lo = (int)((int64_t)count * g / groups), hi = (int)((int64_t)count * (g + 1) / groups);
            slotOf[g] = g; groupFirst[g] = first + lo; groupCount[g] = hi - lo;
        }
    }

    const float inf = std::numeric_limits<float>::infinity();
    float lo[3][8], hi[3][8];
    const uint32_t leafRef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < 8; c++) for (int a = 0; a < 3; a++) { lo[a][c] = inf; hi[a][c] = -inf; }
    uint32_t imask = 0;
    const uint32_t childBase = (uint32_t)(out.nodes.size() / kNodeFloat4s);
    out.nodes.resize(out.nodes.size() + (size_t)groups * kNodeFloat4s);
    for (int g = 0; g < groups; g++) {
        Box b;
        for (int i = 0; i < groupCount[g]; i++) b.grow(items[(size_t)(groupFirst[g] + i)].box);
        for (int a = 0; a < 3; a++) { lo[a][slotOf[g]] = b.lo[a]; hi[a][slotOf[g]] = b.hi[a]; }
        imask |= 1u << slotOf[g];
    }
    encodeNode(&out.nodes[(size_t)self * kNodeFloat4s], lo, hi, leafRef, childBase, imask, 0u, kNoModel);
    // slots ascend with g, so the g-th group is the g-th internal child
    for (int g = 0; g < groups; g++) buildForest(items, groupFirst[g], groupCount[g], childBase + (uint32_t)g, depth + 1, out, rootIndex, deepest);
}

}  // namespace

bool flattenScene(const rt_scene_desc& s, FlatScene& out, std::string& err, bool meshesOnDevice) {
    out = FlatScene();
    if (s.sphereCount < 0 || s.planeCount < 0 || s.AABBCount < 0 || s.modelCount < 0) { err = "negative object count"; return false; }
    if (s.lightCount < 1 || !s.lights) { err = "scene needs at least one light (lights[0] is the only one read)"; return false; }
    if ((s.sphereCount && !s.spheres) || (s.planeCount && !s.planes) || (s.AABBCount && !s.AABBs) || (s.modelCount && !s.models)) {
        err = "null object array with non-zero count"; return false;
    }
    OctreeParams params;
    if (s.octreeMaxDepth > 0) params.maxDepth = s.octreeMaxDepth;
    if (s.octreeLeafTriangles > 0) { params.leafTriangles = s.octreeLeafTriangles; params.sahLeafTriangles = s.octreeLeafTriangles; }
    if (const char* e = std::getenv("RT_WIDEN")) params.widen = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_SAH")) params.sah = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_COLLAPSE")) params.optimalCollapse = std::atoi(e) != 0;
    if (params.maxDepth > kMaxModelOctreeDepth) { err = "octreeMaxDepth exceeds " + std::to_string(kMaxModelOctreeDepth); return false; }

    F4 mat[3];
    for (int i = 0; i < s.sphereCount; i++) {
        const rt_sphere& o = s.spheres[i];
        out.sphGeom.push_back(f4(o.position, o.radius));
        packMaterial(o.mat, o.objectType, (o.debug ? 1u : 0u) | (isIgnoredPosition(o.position) ? 2u : 0u), mat);
        out.sphMat.insert(out.sphMat.end(), mat, mat + 3);
    }
    for (int i = 0; i < s.planeCount; i++) {
        const rt_plane& o = s.planes[i];
        out.plnP.push_back(f4(o.position, 0.0f));
        out.plnN.push_back(f4(o.n, 0.0f));
        {   // Plane.cpp:24 (normal = -n), then kernels.cu:90-93 once per trace routine and kernels.cu:240 before the lighting:
            // normalise = v / sqrt((x x + y y) + z z), IEEE float without contraction, as the kernel evaluates it
            float v[3] = {-o.n.x, -o.n.y, -o.n.z};
            out.plnNormals.push_back(F4{v[0], v[1], v[2], 0.0f});
            for (int k = 0; k < 4; k++) {
                const float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len;
                out.plnNormals.push_back(F4{v[0], v[1], v[2], 0.0f});
            }
        }
        packMaterial(o.mat, o.objectType, (o.debug ? 1u : 0u) | (isIgnoredPosition(o.position) ? 2u : 0u), mat);
        out.plnMat.insert(out.plnMat.end(), mat, mat + 3);
    }
    for (int i = 0; i < s.AABBCount; i++) {
        const rt_aabb& o = s.AABBs[i];
        out.boxMin.push_back(f4(o.min, 0.0f));
        out.boxMax.push_back(f4(o.max, 0.0f));
        packMaterial(o.mat, o.objectType, o.debug ? 1u : 0u, mat);
        out.boxMat.insert(out.boxMat.end(), mat, mat + 3);
    }
    // Forest over the models: reserve a root node for every model with reachable triangles and group those roots
    // by position, so that a ray only meets the models whose (padded) triangle bounds it crosses.
    std::vector<Box> tight((size_t)s.modelCount);
    std::vector<float> pad((size_t)s.modelCount, 0.0f);
    std::vector<uint32_t> rootIndex((size_t)s.modelCount, kNoRoot);
    std::vector<ForestItem> items;
    for (int i = 0; i < s.modelCount; i++) {
        if (!modelBounds(s.models[i], i, tight[(size_t)i], pad[(size_t)i], err)) return false;
        if (reachableTriangles(s.models[i].indicesCount) == 0) continue;
        ForestItem it;
        it.model = i;
        for (int a = 0; a < 3; a++) {
            it.box.lo[a] = tight[(size_t)i].lo[a] - pad[(size_t)i];
            it.box.hi[a] = tight[(size_t)i].hi[a] + pad[(size_t)i];
            it.centre[a] = 0.5f * it.box.lo[a] + 0.5f * it.box.hi[a];
        }
        items.push_back(it);
    }
    int forestDepth = 0;
    if (!items.empty()) {
        out.forestRoot = 0;
        out.nodes.resize(kNodeFloat4s);
        buildForest(items, 0, (int)items.size(), 0u, 0, out, rootIndex, forestDepth);
    }
    if (!meshesOnDevice) {
        int64_t total = 0;
        for (int i = 0; i < s.modelCount; i++) total += reachableTriangles(s.models[i].indicesCount);
        out.attrs.reserve((size_t)total * kAttrFloat4s);
        out.tris.reserve(((size_t)total + kSpareTriangles) * 3);
    }
    std::vector<HostTree> ahead((size_t)s.modelCount);
    if (!meshesOnDevice && !buildTreesAhead(s, params, ahead)) { err = "out of memory while building the octrees"; return false; }
    for (int i = 0; i < s.modelCount; i++) {
        if (!flattenModel(s.models[i], i, params, rootIndex[(size_t)i], tight[(size_t)i], pad[(size_t)i], meshesOnDevice, ahead[(size_t)i], out, err)) return false;
        ahead[(size_t)i] = HostTree();      // the tree is emitted: release it
    }
    out.forestDepth = forestDepth;
    out.params = params;
    out.maxDepth += forestDepth;
    if (out.maxDepth >= kMaxOctreeDepth) { err = "forest plus octree nesting exceeds the depth the traversal reserves pool entries for"; return false; }
    // spare triangle records: the split waves read eight consecutive records wherever a leaf starts
    if (!out.tris.empty()) { for (int k = 0; k < 3 * kSpareTriangles; k++) out.tris.push_back(f4(0.0f, 0.0f, 0.0f, 0.0f)); }
    if (out.triangleCount + kSpareTriangles > (int64_t)kMaxTriangleRecords) { err = "too many triangles in the scene (" + std::to_string(kMaxTriangleRecords) + " records at most)"; return false; }
    out.nodeCount = (int64_t)(out.nodes.size() / kNodeFloat4s);
    out.lightPos[0] = s.lights[0].position.x; out.lightPos[1] = s.lights[0].position.y; out.lightPos[2] = s.lights[0].position.z;
    out.lightCol[0] = s.lights[0].colour.x; out.lightCol[1] = s.lights[0].colour.y; out.lightCol[2] = s.lights[0].colour.z;
    return true;
}

namespace {

struct Walker {
    const rt_scene_desc& scene;
    const FlatScene& flat;
    std::vector<std::vector<uint8_t>> seen;   // per model, per reachable triangle
    std::vector<uint8_t> rootSeen;
    std::string err;
    int deepest = 0;

    bool inside(const double lo[3], const double hi[3], const rt_vec3& p) const {
        return p.x >= lo[0] && p.x <= hi[0] && p.y >= lo[1] && p.y <= hi[1] && p.z >= lo[2] && p.z <= hi[2];
    }

    bool leaf(int m, uint32_t start, const double lo[3], const double hi[3]) {
        const rt_model& model = scene.models[m];
        std::vector<uint8_t>& mine = seen[(size_t)m];
        for (uint32_t t = start;; t++) {
            if ((size_t)t * 3 + 2 >= flat.tris.size()) { err = "leaf runs past the triangle array"; return false; }
            uint32_t idw, owner;
            std::memcpy(&idw, &flat.tris[(size_t)t * 3].w, 4);
            std::memcpy(&owner, &flat.tris[(size_t)t * 3 + 1].w, 4);
            if (owner != (uint32_t)m) {
                err = "triangle record " + std::to_string(t) + " (leaf starting at " + std::to_string(start) + ") of model " + std::to_string(m) +
                      " is tagged with model " + std::to_string(owner);
                return false;
            }
            uint32_t id = idw & kTriIdMask;
            if (id >= mine.size()) { err = "triangle id out of range"; return false; }
            if (mine[id]) { err = "triangle " + std::to_string(id) + " appears in two leaves"; return false; }
            mine[id] = 1;
            const int32_t* ix = model.indices + 3 * (int64_t)id;
            const rt_vec3& p0 = model.vertices[ix[0]].position;
            const F4& v0 = flat.tris[(size_t)t * 3];
            if (v0.x != p0.x || v0.y != p0.y || v0.z != p0.z) { err = "triangle record does not match its vertex"; return false; }
            for (int k = 0; k < 3; k++) {
                if (!inside(lo, hi, model.vertices[ix[k]].position)) { err = "triangle " + std::to_string(id) + " pokes out of its leaf box"; return false; }
            }
            if (idw & kTriLast) return true;
        }
    }

    // m: model whose octree the node belongs to, -1 above the model roots
    bool node(uint32_t index, int m, const double plo[3], const double phi[3], int depth) {
        if (depth >= kMaxOctreeDepth) { err = "nesting deeper than the traversal reserves pool entries for"; return false; }
        deepest = std::max(deepest, depth + 1);
        if ((size_t)(index + 1) * kNodeFloat4s > flat.nodes.size()) { err = "node index out of range"; return false; }
        const F4* rec = &flat.nodes[(size_t)index * kNodeFloat4s];
        uint32_t childBase, masks, owner, expo;
        std::memcpy(&childBase, &rec[1].x, 4); std::memcpy(&masks, &rec[1].y, 4); std::memcpy(&owner, &rec[1].z, 4);
        std::memcpy(&expo, &rec[0].w, 4);
        if (owner != kNoModel) {
            if (m >= 0) { err = "model root below another model's root"; return false; }
            if (owner >= (uint32_t)scene.modelCount) { err = "model root with a model index out of range"; return false; }
            if (rootSeen[owner]) { err = "model " + std::to_string(owner) + " has two roots"; return false; }
            if (flat.mdlInfo[4 * (size_t)owner] != index) { err = "model table does not point at the model's root"; return false; }
            rootSeen[owner] = 1;
            m = (int)owner;
        }
        const uint32_t imask = masks & 0xffu, lmask = (masks >> 8) & 0xffu;
        if (imask & lmask) { err = "slot marked both internal and leaf"; return false; }
        if (m < 0 && lmask) { err = "leaf above the model roots"; return false; }
        const float origin[3] = {rec[0].x, rec[0].y, rec[0].z};
        uint32_t next = childBase;
        for (int c = 0; c < 8; c++) {
            if (!(((imask | lmask) >> c) & 1u)) continue;
            // the slot's box as the traversal decodes it, clipped by the boxes of the slots above: a ray reaches this
            // slot only through all of them, so everything stored below must lie inside their intersection
            double lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                const int e = (int)((expo >> (8 * a)) & 0xffu);
                if (e < 1 || e > 254) { err = "quantisation step out of range"; return false; }
                uint32_t wl, wh;
                std::memcpy(&wl, &(&rec[2 + a].x)[c / 4], 4); std::memcpy(&wh, &(&rec[2 + a].x)[2 + c / 4], 4);
                const double step = std::ldexp(1.0, e - 127);
                const double l = (double)origin[a] + (double)((wl >> (8 * (c % 4))) & 0xffu) * step;
                const double h = (double)origin[a] + (double)((wh >> (8 * (c % 4))) & 0xffu) * step;
                lo[a] = std::max(plo[a], l);
                hi[a] = std::min(phi[a], h);
            }
            if ((lmask >> c) & 1u) {
                uint32_t ref; std::memcpy(&ref, &(&rec[5 + c / 4].x)[c % 4], 4);
                if (!leaf(m, ref, lo, hi)) {
                    err += " [node " + std::to_string(index) + " slot " + std::to_string(c) + "]";
                    return false;
                }
            } else {
                if (!node(next++, m, lo, hi, depth + 1)) {
                    err += " <- node " + std::to_string(index) + " (childBase " + std::to_string(childBase) + ", masks " + std::to_string(masks) + ", slot " + std::to_string(c) + ")";
                    return false;
                }
            }
        }
        return true;
    }
};

}  // namespace

bool validateFlatScene(const rt_scene_desc& scene, const FlatScene& flat, std::string& err) {
    if ((int)flat.mdlInfo.size() != 4 * scene.modelCount) { err = "model table size mismatch"; return false; }
    const double inf = std::numeric_limits<double>::infinity();
    Walker w{scene, flat, {}, std::vector<uint8_t>((size_t)scene.modelCount, 0), std::string(), 0};
    for (int m = 0; m < scene.modelCount; m++) w.seen.emplace_back((size_t)reachableTriangles(scene.models[m].indicesCount), 0);
    const double lo[3] = {-inf, -inf, -inf}, hi[3] = {inf, inf, inf};
    if (flat.forestRoot != kNoRoot && !w.node(flat.forestRoot, -1, lo, hi, 0)) { err = "forest: " + w.err; return false; }
    for (int m = 0; m < scene.modelCount; m++) {
        const bool empty = w.seen[(size_t)m].empty();
        if (empty != (flat.mdlInfo[4 * (size_t)m] == kNoRoot)) { err = "model " + std::to_string(m) + ": root reference does not match its triangle count"; return false; }
        if (!empty && !w.rootSeen[(size_t)m]) { err = "model " + std::to_string(m) + ": the forest never reaches its root"; return false; }
        for (size_t t = 0; t < w.seen[(size_t)m].size(); t++) {
            if (!w.seen[(size_t)m][t]) { err = "model " + std::to_string(m) + ": triangle " + std::to_string(t) + " is in no leaf"; return false; }
        }
    }
    return true;
}

}  // namespace rt
