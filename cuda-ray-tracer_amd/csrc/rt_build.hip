// Octree construction on the GPU.
//
// Produces the same device arrays as the host flattener (rt_scene.cpp) - 128-byte quantised nodes, leaf-ordered
// triangle records, per-triangle shading attributes - for every model at once, from the callers' vertex and index
// arrays.  The reference never builds its octree (Octree.cpp:11-43 is not spatial and has no caller); like the host
// builder this one is an accelerator whose shape does not change any pixel.
//
//   1. one thread per triangle: bounds, centre, 30-bit Morton code of the centre inside the model's bounds;
//      key = (model, code)
//   2. radix sort of (key, triangle)
//   3. top-down, one launch per nesting level, one thread per node: the node's children are the runs of equal
//      3-bit digit at the node's Morton level (found by binary search, levels without a split are skipped), and while
//      slots are free the run with the largest cell is replaced by its own runs; a run with few
//      triangles, identical codes or at the depth cap becomes a leaf, the others get contiguous node slots
//   4. bottom-up, one launch per level: child boxes from the triangles of leaf children and the unions of internal
//      children, padded and quantised exactly as on the host (rt_node_codec.h)
//   5. one thread per triangle: leaf-ordered triangle records and attribute records
#include "rt_build.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_node_codec.h"

namespace rt {

namespace {

constexpr int kMortonLevels = 10;   // 3 bits per level

struct BuildModelDev {
    const rt_vertex* verts;
    const int32_t* indices;
    uint32_t triStart, triCount;
    uint32_t rootIndex;
    int32_t model;
    float lo[3], hi[3];
    float pad;
    int32_t perBoxTris;
    int32_t chunkIndices[8];
};

struct BuildNode {
    uint32_t a, b;          // range of sorted triangle positions
    uint32_t model;         // index into the build-model table
    uint16_t level;         // Morton level whose digit splits the range
    uint16_t depth;         // nesting depth inside the model (root = 0)
};

enum { kCtrNodes = 0, kCtrLeaves, kCtrDepth, kCtrOverflow, kCtrBadNode, kCtrCount };

__device__ __forceinline__ uint32_t expandBits10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ uint32_t modelOfTriangle(const BuildModelDev* models, int count, uint32_t g) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (models[mid].triStart <= g) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

__global__ __launch_bounds__(256) void mortonKernel(const BuildModelDev* __restrict__ models, int modelCount, uint32_t total,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                   F4* __restrict__ triLo, F4* __restrict__ triHi) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const uint32_t mi = modelOfTriangle(models, modelCount, g);
    const BuildModelDev& M = models[mi];
    const int32_t* ix = M.indices + 3 * (size_t)(g - M.triStart);
    const rt_vec3 p0 = M.verts[ix[0]].position, p1 = M.verts[ix[1]].position, p2 = M.verts[ix[2]].position;
    const float lo[3] = {fminf(p0.x, fminf(p1.x, p2.x)), fminf(p0.y, fminf(p1.y, p2.y)), fminf(p0.z, fminf(p1.z, p2.z))};
    const float hi[3] = {fmaxf(p0.x, fmaxf(p1.x, p2.x)), fmaxf(p0.y, fmaxf(p1.y, p2.y)), fmaxf(p0.z, fmaxf(p1.z, p2.z))};
    triLo[g] = F4{lo[0], lo[1], lo[2], 0.0f};
    triHi[g] = F4{hi[0], hi[1], hi[2], 0.0f};
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float centre = 0.5f * lo[a] + 0.5f * hi[a];
        const float extent = M.hi[a] - M.lo[a];
        float f = extent > 0.0f ? (centre - M.lo[a]) / extent * 1024.0f : 0.0f;
        f = fminf(fmaxf(f, 0.0f), 1023.0f);
        q[a] = (uint32_t)f;
    }
    const uint32_t code = expandBits10(q[0]) | (expandBits10(q[1]) << 1) | (expandBits10(q[2]) << 2);
    keys[g] = ((uint64_t)mi << 30) | (uint64_t)code;
    vals[g] = g;
}

__global__ void seedRootsKernel(const BuildModelDev* __restrict__ models, int modelCount, BuildNode* __restrict__ bn,
                                uint32_t* __restrict__ rootList) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= modelCount) return;
    BuildNode n;
    n.a = models[i].triStart; n.b = models[i].triStart + models[i].triCount;
    n.model = (uint32_t)i; n.level = 0; n.depth = 0;
    bn[models[i].rootIndex] = n;
    rootList[i] = models[i].rootIndex;
}

__device__ __forceinline__ uint32_t digitAt(uint64_t key, int level) { return (uint32_t)(key >> (3 * (kMortonLevels - 1 - level))) & 7u; }

// First Morton level >= from at which the sorted range [a, b) splits: its first and last key differ there.
__device__ __forceinline__ int firstSplitLevel(const uint64_t* __restrict__ keys, uint32_t a, uint32_t b, int from) {
    int level = from < kMortonLevels - 1 ? from : kMortonLevels - 1;
    while (level < kMortonLevels - 1 && digitAt(keys[a], level) == digitAt(keys[b - 1], level)) level++;
    return level;
}

// The non-empty runs of equal digit at `level` in the sorted range [a, b), in digit order; returns their number.
__device__ __forceinline__ int cutRuns(const uint64_t* __restrict__ keys, uint32_t a, uint32_t b, int level, uint32_t* runA, uint32_t* runB) {
    int runs = 0;
    uint32_t from = a;
    for (uint32_t c = 0; c < 8 && from < b; c++) {
        // end of the run with digit c = first position whose digit is > c
        uint32_t lo = from, hi = b;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (digitAt(keys[mid], level) <= c) lo = mid + 1; else hi = mid;
        }
        if (lo > from) { runA[runs] = from; runB[runs] = lo; runs++; }
        from = lo;
    }
    return runs;
}

// A run cut at `level` below a node of nesting depth `depth` stays a leaf if it is small, cannot be cut further or would
// nest deeper than the cap.
__device__ __forceinline__ bool isLeafRun(const uint64_t* __restrict__ keys, uint32_t a, uint32_t b, int level, int depth, int leafTriangles, int maxDepth) {
    return (int)(b - a) <= leafTriangles || depth + 1 >= maxDepth || level == kMortonLevels - 1 || keys[a] == keys[b - 1];
}

__device__ __forceinline__ uint32_t compactBits10(uint32_t v) {
    v &= 0x09249249u;
    v = (v | (v >> 2)) & 0x030C30C3u;
    v = (v | (v >> 4)) & 0x0300F00Fu;
    v = (v | (v >> 8)) & 0x030000FFu;
    v = (v | (v >> 16)) & 0x3FFu;
    return v;
}

// Twice the centre (grid units, 1024 per axis) and the edge length of the Morton cell that `key` lies in when the digits
// of levels 0 .. level are fixed; level -1 = the whole grid.
__device__ __forceinline__ void cellCentre2(uint64_t key, int level, int* centre2, int& size) {
    const uint32_t code = (uint32_t)key & 0x3FFFFFFFu;
    const int shift = kMortonLevels - 1 - level;      // 10 for the whole grid
    size = 1 << shift;
    for (int a = 0; a < 3; a++) {
        const int q = (int)compactBits10(code >> a);
        centre2[a] = 2 * ((q >> shift) << shift) + size;
    }
}

// Step 3.  Writes the topology words of each node record (4-7 and 20-27); the boxes follow in boxKernel.
__global__ __launch_bounds__(128) void splitKernel(const uint32_t* __restrict__ nodeList, uint32_t first, uint32_t count,
                                                  BuildNode* __restrict__ bn, const uint64_t* __restrict__ keys, F4* __restrict__ nodes,
                                                  uint8_t* __restrict__ lastFlag, uint32_t* __restrict__ counters, uint32_t capacity,
                                                  int leafTriangles, int maxDepth, bool fold) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t idx = nodeList ? nodeList[i] : first + i;
    const BuildNode n = bn[idx];
    if (n.b <= n.a) { atomicMax(&counters[kCtrBadNode], idx + 1u); return; }   // cannot happen; refuse to read out of bounds
    uint32_t* words = reinterpret_cast<uint32_t*>(nodes + (size_t)idx * kNodeFloat4s);   // topology words: 4-6 meta, 20-27 leaf references
    for (int c = 0; c < 8; c++) words[20 + c] = 0u;
    uint32_t imask = 0, lmask = 0, childBase = 0;
    if (keys[n.a] == keys[n.b - 1]) {
        // nothing to split by (only possible for a model root): one leaf with everything
        lmask = 1u; words[20] = n.a; lastFlag[n.b - 1] = 1;
        atomicAdd(&counters[kCtrLeaves], 1u);
        atomicMax(&counters[kCtrDepth], (uint32_t)n.depth + 1u);
    } else {
        // first Morton level at which the range splits: the first and last key differ there (the keys are sorted and agree on
        // every level above the node's own)
        const int level = firstSplitLevel(keys, n.a, n.b, n.level);
        // The children: runs of equal digit at that level.  An octree of a surface uses about half of a node's eight slots;
        // as long as there is room, the internal child with the largest cell (then the most triangles) is replaced by its
        // own runs, so that a ray visits fewer nodes for the same triangles.
        uint32_t kidA[8], kidB[8];
        int kidLevel[8];                      // Morton level the kid's run was cut at
        bool kidLeaf[8], kidTried[8];
        int kids = 0;
        {
            uint32_t runA[8], runB[8];
            const int runs = cutRuns(keys, n.a, n.b, level, runA, runB);
            for (int r = 0; r < runs; r++) {
                kidA[kids] = runA[r]; kidB[kids] = runB[r]; kidLevel[kids] = level; kidTried[kids] = false;
                kidLeaf[kids] = isLeafRun(keys, runA[r], runB[r], level, n.depth, leafTriangles, maxDepth);
                kids++;
            }
        }
        while (fold && kids < 8) {
            int pick = -1;
            for (int k = 0; k < kids; k++) {
                if (kidLeaf[k] || kidTried[k]) continue;
                if (pick < 0 || kidLevel[k] < kidLevel[pick] || (kidLevel[k] == kidLevel[pick] && kidB[k] - kidA[k] > kidB[pick] - kidA[pick])) pick = k;
            }
            if (pick < 0) break;
            kidTried[pick] = true;
            const int sub = firstSplitLevel(keys, kidA[pick], kidB[pick], kidLevel[pick] + 1);
            uint32_t runA[8], runB[8];
            const int runs = cutRuns(keys, kidA[pick], kidB[pick], sub, runA, runB);
            if (kids - 1 + runs > 8) continue;
            for (int r = 0; r < runs; r++) {
                const int k = r == 0 ? pick : kids++;
                kidA[k] = runA[r]; kidB[k] = runB[r]; kidLevel[k] = sub; kidTried[k] = false;
                kidLeaf[k] = isLeafRun(keys, runA[r], runB[r], sub, n.depth, leafTriangles, maxDepth);
            }
        }
        // Slots: bit a of a slot index means "towards +a within the node", which the traversal's front-to-back order
        // (slot ^ ray octant) relies on; the order is a heuristic, hits do not depend on it.  Greedily the best total of
        // (centre of the kid's cell - centre of the node's cell) . (slot direction); unfolded kids get their own digit.
        int slotOfKid[8];
        {
            int nodeCentre[3], nodeSize;
            cellCentre2(keys[n.a], level - 1, nodeCentre, nodeSize);
            float score[8][8];
            for (int k = 0; k < kids; k++) {
                int c[3], size;
                cellCentre2(keys[kidA[k]], kidLevel[k], c, size);
                float rel[3];
                for (int ax = 0; ax < 3; ax++) rel[ax] = (float)(c[ax] - nodeCentre[ax]) / (float)(2 * nodeSize);
                for (int sl = 0; sl < 8; sl++)
                    score[k][sl] = ((sl & 1) ? rel[0] : -rel[0]) + ((sl & 2) ? rel[1] : -rel[1]) + ((sl & 4) ? rel[2] : -rel[2]);
                slotOfKid[k] = -1;
            }
            uint32_t slotTaken = 0;
            for (int round = 0; round < kids; round++) {
                int bk = -1, bs = -1;
                for (int k = 0; k < kids; k++) {
                    if (slotOfKid[k] >= 0) continue;
                    for (int sl = 0; sl < 8; sl++) {
                        if ((slotTaken >> sl) & 1u) continue;
                        if (bk < 0 || score[k][sl] > score[bk][bs]) { bk = k; bs = sl; }
                    }
                }
                slotOfKid[bk] = bs; slotTaken |= 1u << bs;
            }
        }
        int kidOfSlot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        for (int k = 0; k < kids; k++) kidOfSlot[slotOfKid[k]] = k;
        uint32_t internal = 0;
        for (int c = 0; c < 8; c++) {
            const int k = kidOfSlot[c];
            if (k < 0) continue;
            if (kidLeaf[k]) { lmask |= 1u << c; words[20 + c] = kidA[k]; lastFlag[kidB[k] - 1] = 1; }
            else { imask |= 1u << c; internal++; }
        }
        atomicAdd(&counters[kCtrLeaves], (uint32_t)__builtin_popcount(lmask));
        if (internal) {
            childBase = atomicAdd(&counters[kCtrNodes], internal);
            if (childBase + internal > capacity) { atomicExch(&counters[kCtrOverflow], 1u); imask = 0; }
            uint32_t next = childBase;
            for (int c = 0; c < 8; c++) {
                if (!((imask >> c) & 1u)) continue;
                const int k = kidOfSlot[c];
                BuildNode ch;
                ch.a = kidA[k]; ch.b = kidB[k]; ch.model = n.model; ch.level = (uint16_t)(kidLevel[k] + 1); ch.depth = (uint16_t)(n.depth + 1);
                bn[next++] = ch;
            }
            atomicMax(&counters[kCtrDepth], (uint32_t)n.depth + 2u);
        } else {
            atomicMax(&counters[kCtrDepth], (uint32_t)n.depth + 1u);
        }
    }
    words[4] = childBase; words[5] = imask | (lmask << 8); words[6] = kNoModel; words[7] = 0u;
}

// Step 4 for the nodes of one level (deeper levels are already done).
__global__ __launch_bounds__(128) void boxKernel(const uint32_t* __restrict__ nodeList, uint32_t first, uint32_t count,
                                                const BuildNode* __restrict__ bn, const BuildModelDev* __restrict__ models,
                                                const uint32_t* __restrict__ vals, const uint8_t* __restrict__ lastFlag,
                                                const F4* __restrict__ triLo, const F4* __restrict__ triHi, F4* __restrict__ nodes,
                                                F4* __restrict__ unionLo, F4* __restrict__ unionHi, uint32_t* __restrict__ dbg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t idx = nodeList ? nodeList[i] : first + i;
    const BuildNode n = bn[idx];
    const BuildModelDev& M = models[n.model];
    uint4* rec = reinterpret_cast<uint4*>(nodes + (size_t)idx * kNodeFloat4s);
    const uint4 meta = rec[1], refA = rec[5], refB = rec[6];
    const uint32_t childBase = meta.x, masks = meta.y;
    const uint32_t imask = masks & 0xffu, lmask = (masks >> 8) & 0xffu;
    const uint32_t leafRef[8] = {refA.x, refA.y, refA.z, refA.w, refB.x, refB.y, refB.z, refB.w};
    const float inf = __builtin_inff();
    float lo[3][8], hi[3][8];
    float ulo[3] = {inf, inf, inf}, uhi[3] = {-inf, -inf, -inf};
    uint32_t next = childBase;
    for (int c = 0; c < 8; c++) {
        float blo[3] = {inf, inf, inf}, bhi[3] = {-inf, -inf, -inf};
        if ((lmask >> c) & 1u) {
            if (leafRef[c] < n.a || leafRef[c] >= n.b) { atomicMax(&dbg[0], idx + 1u); continue; }   // never: keeps the loop below in bounds
            for (uint32_t p = leafRef[c];; p++) {
                const F4 l = triLo[vals[p]], h = triHi[vals[p]];
                blo[0] = fminf(blo[0], l.x); blo[1] = fminf(blo[1], l.y); blo[2] = fminf(blo[2], l.z);
                bhi[0] = fmaxf(bhi[0], h.x); bhi[1] = fmaxf(bhi[1], h.y); bhi[2] = fmaxf(bhi[2], h.z);
                if (lastFlag[p]) break;
            }
        } else if ((imask >> c) & 1u) {
            if (bn[next].model != n.model || bn[next].a < n.a || bn[next].b > n.b) atomicMax(&dbg[0], idx + 1u);
            const F4 l = unionLo[next], h = unionHi[next];
            next++;
            blo[0] = l.x; blo[1] = l.y; blo[2] = l.z; bhi[0] = h.x; bhi[1] = h.y; bhi[2] = h.z;
        }
        for (int a = 0; a < 3; a++) {
            lo[a][c] = blo[a] - M.pad; hi[a][c] = bhi[a] + M.pad;
            ulo[a] = fminf(ulo[a], blo[a]); uhi[a] = fmaxf(uhi[a], bhi[a]);
        }
    }
    unionLo[idx] = F4{ulo[0], ulo[1], ulo[2], 0.0f};
    unionHi[idx] = F4{uhi[0], uhi[1], uhi[2], 0.0f};
    uint32_t words[32];
    encodeNodeWords(words, lo, hi, leafRef, childBase, imask, lmask, n.depth == 0 ? (uint32_t)M.model : kNoModel);
    for (int k = 0; k < kNodeFloat4s; k++) rec[k] = make_uint4(words[4 * k], words[4 * k + 1], words[4 * k + 2], words[4 * k + 3]);
}

__global__ __launch_bounds__(256) void emitTrianglesKernel(const BuildModelDev* __restrict__ models, int modelCount, uint32_t total,
                                                          const uint32_t* __restrict__ vals, const uint8_t* __restrict__ lastFlag,
                                                          F4* __restrict__ tris) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t g = vals[p];
    const BuildModelDev& M = models[modelOfTriangle(models, modelCount, g)];
    const uint32_t t = g - M.triStart;
    const int32_t* ix = M.indices + 3 * (size_t)t;
    const rt_vec3 p0 = M.verts[ix[0]].position, p1 = M.verts[ix[1]].position, p2 = M.verts[ix[2]].position;
    const uint32_t id = t | chunkTagOf((int64_t)t, (int64_t)M.perBoxTris, M.chunkIndices) | (lastFlag[p] ? kTriLast : 0u);
    tris[3 * (size_t)p + 0] = F4{p0.x, p0.y, p0.z, codecBitsToFloat(id)};
    tris[3 * (size_t)p + 1] = F4{p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, codecBitsToFloat((uint32_t)M.model)};   // Model.cpp:137
    tris[3 * (size_t)p + 2] = F4{p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, codecBitsToFloat(M.triStart)};          // Model.cpp:138
}

__global__ __launch_bounds__(256) void emitAttributesKernel(const BuildModelDev* __restrict__ models, int modelCount, uint32_t total,
                                                           F4* __restrict__ attrs) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const BuildModelDev& M = models[modelOfTriangle(models, modelCount, g)];
    const int32_t* ix = M.indices + 3 * (size_t)(g - M.triStart);
    const rt_vertex& a = M.verts[ix[0]];
    const rt_vertex& b = M.verts[ix[1]];
    const rt_vertex& c = M.verts[ix[2]];
    F4* rec = attrs + (size_t)g * kAttrFloat4s;
    rec[0] = F4{a.normal.x, a.normal.y, a.normal.z, a.ambient.x};
    rec[1] = F4{b.normal.x, b.normal.y, b.normal.z, a.ambient.y};
    rec[2] = F4{c.normal.x, c.normal.y, c.normal.z, a.ambient.z};
    rec[3] = F4{b.ambient.x, b.ambient.y, b.ambient.z, a.diffuse.x};
    rec[4] = F4{c.ambient.x, c.ambient.y, c.ambient.z, a.diffuse.y};
    rec[5] = F4{b.diffuse.x, b.diffuse.y, b.diffuse.z, a.diffuse.z};
    rec[6] = F4{c.diffuse.x, c.diffuse.y, c.diffuse.z, 0.0f};
}

// Owns the temporary device allocations of one build.
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) { ptrs.push_back(p); *out = static_cast<T*>(p); }
        return e;
    }
};

}  // namespace

#define RT_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return e_; } } while (0)

hipError_t buildMeshesOnDevice(const rt_scene_desc& scene, const FlatScene& flat, DeviceMeshes& out, std::string& err) {
    out = DeviceMeshes();
    const int modelCount = (int)flat.deviceModels.size();
    const uint32_t total = (uint32_t)flat.triangleCount;
    if (modelCount == 0 || total == 0) return hipSuccess;
    if (flat.triangleCount + kSpareTriangles > (int64_t)kMaxTriangleRecords) { err = "too many triangles for the device builder"; return hipErrorInvalidValue; }
    if (modelCount >= (1 << 24)) { err = "too many models for the device builder"; return hipErrorInvalidValue; }
    const uint32_t forestNodes = (uint32_t)(flat.nodes.size() / kNodeFloat4s);
    const uint32_t capacity = forestNodes + total;     // an internal node below a root has at least two children

    Scratch tmp;
    // callers' vertex and index arrays, one device buffer each
    size_t vertTotal = 0, indexTotal = 0;
    for (const DeviceBuildModel& dm : flat.deviceModels) {
        vertTotal += (size_t)scene.models[dm.model].vertCount;
        indexTotal += 3 * (size_t)dm.triCount;
    }
    rt_vertex* dVerts = nullptr; int32_t* dIndices = nullptr;
    RT_TRY(tmp.get(&dVerts, vertTotal));
    RT_TRY(tmp.get(&dIndices, indexTotal));
    std::vector<BuildModelDev> table((size_t)modelCount);
    size_t vertAt = 0, indexAt = 0;
    for (int i = 0; i < modelCount; i++) {
        const DeviceBuildModel& dm = flat.deviceModels[(size_t)i];
        const rt_model& m = scene.models[dm.model];
        RT_TRY(hipMemcpy(dVerts + vertAt, m.vertices, (size_t)m.vertCount * sizeof(rt_vertex), hipMemcpyHostToDevice));
        RT_TRY(hipMemcpy(dIndices + indexAt, m.indices, 3 * (size_t)dm.triCount * sizeof(int32_t), hipMemcpyHostToDevice));
        BuildModelDev& b = table[(size_t)i];
        b.verts = dVerts + vertAt; b.indices = dIndices + indexAt;
        b.triStart = dm.triStart; b.triCount = dm.triCount; b.rootIndex = dm.rootIndex; b.model = dm.model;
        std::memcpy(b.lo, dm.lo, sizeof b.lo); std::memcpy(b.hi, dm.hi, sizeof b.hi);
        b.pad = dm.pad; b.perBoxTris = dm.perBoxTris;
        std::memcpy(b.chunkIndices, dm.chunkIndices, sizeof b.chunkIndices);
        vertAt += (size_t)m.vertCount; indexAt += 3 * (size_t)dm.triCount;
    }
    BuildModelDev* dModels = nullptr;
    RT_TRY(tmp.get(&dModels, (size_t)modelCount));
    RT_TRY(hipMemcpy(dModels, table.data(), table.size() * sizeof(BuildModelDev), hipMemcpyHostToDevice));

    uint64_t *keysIn = nullptr, *keysOut = nullptr; uint32_t *valsIn = nullptr, *valsOut = nullptr;
    F4 *triLo = nullptr, *triHi = nullptr, *unionLo = nullptr, *unionHi = nullptr, *nodes = nullptr;
    uint8_t* lastFlag = nullptr; BuildNode* bn = nullptr; uint32_t *counters = nullptr, *rootList = nullptr;
    RT_TRY(tmp.get(&keysIn, total)); RT_TRY(tmp.get(&keysOut, total));
    RT_TRY(tmp.get(&valsIn, total)); RT_TRY(tmp.get(&valsOut, total));
    RT_TRY(tmp.get(&triLo, total)); RT_TRY(tmp.get(&triHi, total));
    RT_TRY(tmp.get(&lastFlag, total));
    RT_TRY(tmp.get(&bn, capacity)); RT_TRY(tmp.get(&unionLo, capacity)); RT_TRY(tmp.get(&unionHi, capacity));
    RT_TRY(tmp.get(&nodes, (size_t)capacity * kNodeFloat4s));
    RT_TRY(tmp.get(&counters, (size_t)kCtrCount)); RT_TRY(tmp.get(&rootList, (size_t)modelCount));
    RT_TRY(hipMemset(lastFlag, 0, total));
    RT_TRY(hipMemcpy(nodes, flat.nodes.data(), flat.nodes.size() * sizeof(F4), hipMemcpyHostToDevice));   // the forest above the roots
    uint32_t counterInit[kCtrCount] = {};
    counterInit[kCtrNodes] = forestNodes;
    RT_TRY(hipMemcpy(counters, counterInit, sizeof counterInit, hipMemcpyHostToDevice));

    // 1 + 2: codes, sort
    hipLaunchKernelGGL(mortonKernel, dim3((total + 255) / 256), dim3(256), 0, nullptr, dModels, modelCount, total, keysIn, valsIn, triLo, triHi);
    RT_TRY(hipGetLastError());
    int modelBits = 1;
    while ((1 << modelBits) < modelCount) modelBits++;
    size_t sortBytes = 0;
    RT_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, keysIn, keysOut, valsIn, valsOut, (int)total, 0, 30 + modelBits));
    uint8_t* sortTemp = nullptr;
    RT_TRY(tmp.get(&sortTemp, sortBytes));
    RT_TRY(hipcub::DeviceRadixSort::SortPairs(sortTemp, sortBytes, keysIn, keysOut, valsIn, valsOut, (int)total, 0, 30 + modelBits));

    // 3: top-down, level by level; the nodes created by one launch occupy one contiguous index range
    hipLaunchKernelGGL(seedRootsKernel, dim3((modelCount + 127) / 128), dim3(128), 0, nullptr, dModels, modelCount, bn, rootList);
    RT_TRY(hipGetLastError());
    struct Range { uint32_t first, count; };
    std::vector<Range> levels;
    uint32_t done = forestNodes;
    const int maxDepth = flat.params.maxDepth, leafTriangles = flat.params.leafTriangles;
    static const bool fold = !std::getenv("RT_DEVICE_FOLD") || std::atoi(std::getenv("RT_DEVICE_FOLD")) != 0;    // experiment switch: 0 = one Morton digit per node
    hipLaunchKernelGGL(splitKernel, dim3((modelCount + 127) / 128), dim3(128), 0, nullptr, rootList, 0u, (uint32_t)modelCount, bn, keysOut, nodes,
                       lastFlag, counters, capacity, leafTriangles, maxDepth, fold);
    RT_TRY(hipGetLastError());
    uint32_t host[kCtrCount];
    for (;;) {
        RT_TRY(hipMemcpy(host, counters, sizeof host, hipMemcpyDeviceToHost));
        if (host[kCtrOverflow]) { err = "device octree builder ran out of node slots"; return hipErrorOutOfMemory; }
        if (host[kCtrBadNode]) { err = "device octree builder met an empty node range (node " + std::to_string(host[kCtrBadNode] - 1) + ")"; return hipErrorUnknown; }
        const uint32_t created = host[kCtrNodes] - done;
        if (created == 0) break;
        levels.push_back(Range{done, created});
        hipLaunchKernelGGL(splitKernel, dim3((created + 127) / 128), dim3(128), 0, nullptr, (const uint32_t*)nullptr, done, created, bn, keysOut, nodes,
                           lastFlag, counters, capacity, leafTriangles, maxDepth, fold);
        RT_TRY(hipGetLastError());
        done += created;
    }
    const uint32_t nodeCount = done;

    // 4: bottom-up boxes
    for (size_t k = levels.size(); k-- > 0;) {
        hipLaunchKernelGGL(boxKernel, dim3((levels[k].count + 127) / 128), dim3(128), 0, nullptr, (const uint32_t*)nullptr, levels[k].first,
                           levels[k].count, bn, dModels, valsOut, lastFlag, triLo, triHi, nodes, unionLo, unionHi, counters + kCtrBadNode);
        RT_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(boxKernel, dim3((modelCount + 127) / 128), dim3(128), 0, nullptr, rootList, 0u, (uint32_t)modelCount, bn, dModels, valsOut,
                       lastFlag, triLo, triHi, nodes, unionLo, unionHi, counters + kCtrBadNode);
    RT_TRY(hipGetLastError());
    RT_TRY(hipMemcpy(host, counters, sizeof host, hipMemcpyDeviceToHost));
    if (host[kCtrBadNode]) { err = "device octree builder: node " + std::to_string(host[kCtrBadNode] - 1) + " has a child or leaf outside its range"; return hipErrorUnknown; }

    // 5: final arrays (exact sizes; spare triangle records at the end, see kSpareTriangles)
    void* p = nullptr;
    RT_TRY(hipMalloc(&p, (size_t)nodeCount * kNodeFloat4s * sizeof(F4)));
    out.nodes = static_cast<F4*>(p);
    hipError_t e = hipMalloc(&p, ((size_t)total + kSpareTriangles) * 3 * sizeof(F4));
    if (e == hipSuccess) { out.tris = static_cast<F4*>(p); e = hipMalloc(&p, (size_t)total * kAttrFloat4s * sizeof(F4)); }
    if (e == hipSuccess) out.attrs = static_cast<F4*>(p);
    if (e == hipSuccess) e = hipMemcpy(out.nodes, nodes, (size_t)nodeCount * kNodeFloat4s * sizeof(F4), hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(out.tris + 3 * (size_t)total, 0, 3 * kSpareTriangles * sizeof(F4));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(emitTrianglesKernel, dim3((total + 255) / 256), dim3(256), 0, nullptr, dModels, modelCount, total, valsOut, lastFlag, out.tris);
        hipLaunchKernelGGL(emitAttributesKernel, dim3((total + 255) / 256), dim3(256), 0, nullptr, dModels, modelCount, total, out.attrs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        err = std::string("device octree build: ") + hipGetErrorString(e);
        freeDeviceMeshes(out);
        return e;
    }
    out.nodeCount = nodeCount;
    out.leafCount = host[kCtrLeaves];
    out.depth = (int)host[kCtrDepth];
    out.triangleRecords = (int64_t)total + kSpareTriangles;
    return hipSuccess;
}

void freeDeviceMeshes(DeviceMeshes& m) {
    if (m.nodes) (void)hipFree(m.nodes);
    if (m.tris) (void)hipFree(m.tris);
    if (m.attrs) (void)hipFree(m.attrs);
    m = DeviceMeshes();
}

}  // namespace rt
