// Device-side data layout of a flattened scene, shared by the host flattener and the gfx950 kernels.
//
// Everything a ray touches lives in HBM as 16-byte-aligned float4 / uint4 arrays (struct-of-arrays: one array
// per field group), so that a lane fetches a record with a few dwordx4 loads and uniform loops over the small
// analytic-object tables turn into scalar loads.
#pragma once
#include <stdint.h>

namespace rt {

// ---- octree ----
// One node is 128 bytes (one cache line) and describes its eight child octants with boxes quantised to 8 bits per
// plane against a per-node origin and per-axis power-of-two step:
//   u4[0]  { bits(origin.x), bits(origin.y), bits(origin.z), ex | ey << 8 | ez << 16 }   step_a = 2^(e_a - 127)
//   u4[1]  { childBase, internalMask | leafMask << 8, model, 0 }
//   u4[2]  x planes: qlo of children 0-3, qlo of 4-7, qhi of 0-3, qhi of 4-7 (one byte per child, child c in byte c % 4)
//   u4[3]  y planes, u4[4] z planes, same order
//   u4[5..6] leafRef[8]: first triangle of a leaf child in the leaf-ordered triangle array
//   u4[7]  unused
// Child c spans [origin + qlo * step, origin + qhi * step] on each axis; the host rounds qlo down and qhi up, so the
// decoded box always encloses the (already padded) box it was made from.  Empty slots hold qlo = 255, qhi = 0 and are
// masked out by internalMask / leafMask.
// Internal children are stored contiguously from childBase in slot order, so child slot c is node
// childBase + popcount(internalMask & ((1 << c) - 1)).
// Slot c is the octant (x > cx) | (y > cy) << 1 | (z > cz) << 2 of the node's split point.
//
// All models hang off one forest: nodes above the models group model roots by position, and the root node of a model's
// own octree carries the model's index in `model` (kNoModel everywhere else).  The traversal evaluates the model's
// exact bounding-box gate when it reaches such a node.  A scene with a single non-empty model has that model's root as
// the forest root.
constexpr int kNodeFloat4s = 8;
constexpr uint32_t kNoRoot = 0xFFFFFFFFu;     // model without triangles / scene without mesh triangles
constexpr uint32_t kNoModel = 0xFFFFFFFFu;
constexpr int kMaxOctreeDepth = 16;           // nesting levels (forest + model octree); the traversal keeps pool entries in reserve for a depth-first descent of this many levels
constexpr int kMaxModelOctreeDepth = 10;      // cap on the per-model build depth
#ifndef RT_PREGATED_MODELS
#define RT_PREGATED_MODELS 64
#endif
constexpr int kPregatedModels = RT_PREGATED_MODELS;           // up to this many models the gates are evaluated once per ray, up front

// ---- triangles ----
// Leaf-ordered, 3 float4 per triangle: { v0.xyz, bits(id) }, { e1.xyz, bits(model) }, { e2.xyz, bits(attribute base of the model) } with e1 = v1 - v0 and
// e2 = v2 - v0 computed in float on the host (the same subtraction the intersection routine starts with).
// The id word: bits 0-26 triangle index inside its model, bits 27-29 which of the model's eight sequential chunk boxes
// the triangle belongs to, bit 30 set if it sits in the last slot of its chunk (which the eight-box mesh mode never
// tests), bit 31 marks the last triangle of a leaf.
constexpr uint32_t kTriLast = 0x80000000u;
constexpr uint32_t kTriChunkDropped = 0x40000000u;
constexpr int kTriChunkShift = 27;
constexpr uint32_t kTriIdMask = (1u << kTriChunkShift) - 1u;
// Attributes for shading, 7 float4 per triangle, indexed by triBase(model) + id:
//   { n0.xyz, ka0.x } { n1.xyz, ka0.y } { n2.xyz, ka0.z } { ka1.xyz, kd0.x } { ka2.xyz, kd0.y } { kd1.xyz, kd0.z } { kd2.xyz, 0 }
constexpr int kAttrFloat4s = 7;
// The triangle array ends with this many zeroed records: a split wave reads eight consecutive records wherever a leaf starts.
constexpr int kSpareTriangles = 8;
// The traversal queues leaves as (first triangle record | chunk mask << 24).
constexpr uint32_t kMaxTriangleRecords = 1u << 24;

struct F4 { float x, y, z, w; };

// Pointers + counts handed to every kernel by value.
struct DevScene {
    // spheres: geom = { centre.xyz, radius }
    const F4* sphGeom;
    // materials, three float4 per object of any kind:
    //   m0 = { ambientColour.rgb, ambient }  m1 = { diffuseColour.rgb, diffuse }
    //   m2 = { specular, shininess, bits(objectType), bits(flags) }   flags: 1 = debug, 2 = skipped by the ignore test
    const F4* sphMat;
    // planes: p = { position.xyz, 0 }, n = { n.xyz, 0 }
    const F4* plnP;
    const F4* plnN;
    const F4* plnMat;
    // five per plane: -n, and -n normalised once, twice, three and four times (what the reference holds as the hit normal after
    // one to three of its trace routines, and what it hands to the lighting after one more normalisation) - the same for every
    // pixel of a plane, so the host works them out once with the kernel's arithmetic
    const F4* plnNormals;
    // boxes
    const F4* boxMin;
    const F4* boxMax;
    const F4* boxMat;
    // models: gate box (exact reference semantics), tight box of the triangles, material, info
    const F4* mdlGateMin;
    const F4* mdlGateMax;
    const F4* mdlTightMin;     // w: a ray that meets the tight box and whose largest origin coordinate is at most this passes the gate
    const F4* mdlTightMax;     //    (the gate box surrounds the tight box by a margin that outweighs the rounding of both tests; -inf: never)
    const F4* mdlMat;          // same 3-float4 material layout
    const F4* mdlChunkMin;     // 8 per model: the eight sequential-chunk boxes (eight-box mesh mode)
    const F4* mdlChunkMax;
    const uint32_t* mdlInfo;   // 4 per model: { root node (kNoRoot if empty), triBase, 0, 0 }
    const F4* nodes;
    const F4* tris;
    const F4* attrs;
    int32_t nSph, nPln, nBox, nMdl;
    uint32_t forestRoot;       // node the mesh traversal starts from, kNoRoot if no model has a reachable triangle
    float lightPos[3];
    float lightCol[3];
};

// How the tiles of a frame (row-major over the tile grid) are dealt to the ranks of a sharded run: in periods of
// rank0Slots + (world - 1) * otherSlots consecutive tiles, of which the first rank0Slots belong to rank 0, the next otherSlots
// to rank 1, and so on.  1 / 1: tile i belongs to rank i % world.  Rank 0 also reassembles every frame; with fewer slots than
// the others it renders less, by the time that takes.
struct TileDeal {
    int32_t world, rank0Slots, otherSlots;
};
#if defined(__HIPCC__)
#define RT_HOST_DEVICE __host__ __device__
#else
#define RT_HOST_DEVICE
#endif
RT_HOST_DEVICE inline int dealPeriod(TileDeal d) { return d.rank0Slots + (d.world - 1) * d.otherSlots; }
RT_HOST_DEVICE inline int dealFirstSlot(TileDeal d, int rank) { return rank == 0 ? 0 : d.rank0Slots + (rank - 1) * d.otherSlots; }
RT_HOST_DEVICE inline int dealRun(TileDeal d, int rank) { return rank == 0 ? d.rank0Slots : d.otherSlots; }
// tiles of a frame of `tiles` tiles that belong to `rank`: its local tiles 0 .. count - 1, in frame order
RT_HOST_DEVICE inline int dealOwned(TileDeal d, int rank, int tiles) {
    const int period = dealPeriod(d), run = dealRun(d, rank), rest = tiles % period - dealFirstSlot(d, rank);
    return (tiles / period) * run + (rest < 0 ? 0 : (rest > run ? run : rest));
}
RT_HOST_DEVICE inline int dealMostOwned(TileDeal d, int tiles) {
    const int a = dealOwned(d, 0, tiles), b = d.world > 1 ? dealOwned(d, 1, tiles) : 0;     // ranks 1.. own no more than rank 1
    return a > b ? a : b;
}
// which rank a tile belongs to, and which of that rank's tiles it is
RT_HOST_DEVICE inline void dealLocate(TileDeal d, uint32_t tile, uint32_t& rank, uint32_t& local) {
    const uint32_t period = (uint32_t)dealPeriod(d), p = tile / period, s = tile - p * period;
    if (s < (uint32_t)d.rank0Slots) { rank = 0; local = p * (uint32_t)d.rank0Slots + s; return; }
    const uint32_t q = s - (uint32_t)d.rank0Slots, r = q / (uint32_t)d.otherSlots;
    rank = 1u + r; local = p * (uint32_t)d.otherSlots + (q - r * (uint32_t)d.otherSlots);
}

// Per-frame state handed to the render kernel by value.
struct FrameParams {
    int32_t width, height;
    // tile sharding: of every `world` consecutive tiles (a period of the deal, see TileDeal) this launch renders the `tileRun`
    // tiles that start at slot `rank`; plain interleaving: rank = the rank, world = the number of ranks, tileRun = 1.
    // tilesX/tilesY = tile grid
    int32_t rank, world, tileRun, tilesX, tilesY, ownedTiles;
    int32_t compactOutput;     // 1: write tile-major into the rank's compact buffer, 0: row-major full frame
    int32_t blockOrder;        // workgroup -> tile mapping: 0 tile by tile, 2 tiles interleaved over the XCDs
    int32_t nodePoolEntries;   // low half: capacity of the traversal's pool of pending children that a wave may use (96 .. the compiled size), high half: of its pool of triangle work (72 .. compiled size); 0 = compiled size, smaller values only serve tests
    int32_t traversal;         // kTrav* bits; bits 8-15: sibling expansion applies while the pool holds at most 8 x this many entries
    int32_t leadWaves;         // with split: this many regular waves (a multiple of 8) come before the split waves in the grid
    int32_t split;             // 1: the grid starts with kMaxExtraWaves split waves for the heaviest blocks (BlockFeedback::extraMap)
    float camPos[3], camUp[3], camRight[3], camDir[3], fovFactor;
    float eyeDir[3];           // camPos / length(camPos) in float, component by component (the view vector of the specular term)
    // SceneConfig
    float shadowBias, shadowIntensity;
    float sphereStrength, planeStrength, boxStrength, modelStrength;
    float background[3];
    float softRadius;
    int32_t softNum, maxDepth;
    int32_t meshMode;          // kMesh*
    uint32_t flags;            // kCfg* bits
};

enum : uint32_t {
    kCfgHardShadows = 1u << 0, kCfgSoftShadows = 1u << 1, kCfgAreaSpecular = 1u << 2, kCfgReflections = 1u << 3,
    kCfgReflectPlanes = 1u << 4, kCfgRenderBoxes = 1u << 5, kCfgRenderModels = 1u << 6, kCfgAmbient = 1u << 7,
    kCfgDiffuse = 1u << 8, kCfgSpecular = 1u << 9, kCfgAntiAlias = 1u << 10, kCfgCull = 1u << 11,
    kCfgModelsVisible = 1u << 12,  // false when the selected mesh mode renders nothing
    // instrumented kernel only: every closest-hit mesh trace is repeated with the hit distance it found as the bound from the
    // start, and what that second pass visits is counted as the trace's useful work (kCntUseful*)
    kCfgCountUseful = 1u << 13
};

enum : uint32_t { kTravDirectStart = 1u << 0, kTravExpandSiblings = 1u << 1 };

// Mesh hit modes of Model::hit (Model.cpp:161-357) that render something.
enum : int32_t {
    kMeshWholeBox = 0,   // whole-model box gate + every triangle, t > 0 (Model.cpp:275-317)
    kMeshEightBoxes = 1, // eight chunk-box gates, last slot of each chunk skipped, t > 0 (Model.cpp:220-274)
    kMeshNoBox = 2       // no gate and no t > 0 test (Model.cpp:318-356)
};

// Feedback from one frame to the next of the same geometry.  The frame is as long as its slowest wave, and a handful of
// 8x8 pixel blocks (reflective silhouettes) run ten times longer than the rest: every block records its cost, a small
// kernel compares each block with the time the whole frame would need if its work were spread evenly, and the next
// frame renders the blocks that are worth it with eight waves each (one per pixel row: eight rays, and 56 lanes that only help with the traversal), started
// before everything else.
struct BlockFeedback {
    uint32_t* blockCost;       // per block (tile * blocks per tile + block in tile): run time of its wave in the frame being rendered, 10 ns units
    const uint32_t* extraMap;  // block | pixel row << 20 for each split wave
    const uint32_t* extraCount;// number of valid extraMap entries (may exceed kMaxExtraWaves: clamp)
    uint32_t* hint;            // host-visible copy of *extraCount, refreshed by every frame (may be null)
    unsigned long long* frameStart; // when the frame's first workgroup started, 100 MHz clock (may be null)
    const uint32_t* heavyBits; // one bit per block: set = rendered by split waves, the regular wave skips it
};
constexpr int kMaxExtraWaves = 4096;
constexpr int kMaxSplitBlocks = 1 << 20;    // frames with more blocks per rank are rendered without splitting

// Device-side state of the block selection; two slots used alternately by consecutive frames.
constexpr int kCostBins = 128;              // four bins per power of two of a block's cost
struct SplitState {
    unsigned long long total[2];
    unsigned long long binSum[2][kCostBins];
    uint32_t maxCost[2];
    uint32_t used[2];                       // split waves chosen by the selection of that slot
    unsigned long long frameStart;          // start of the latest frame, 100 MHz clock
};

// Per-launch counters (one 64-bit word each), accumulated with one atomic per wave.  The buffer holds kCounterSlots sets of
// them, one cache line apart; workgroup b adds to set b % kCounterSlots and the host adds the sets up.  (Thirty thousand
// atomics on one address from eight XCDs take 0.26 ms, longer than the 1080p frame they count.)
enum { kCntRays = 0, kCntNodeTests, kCntTriTests, kCntAttrFetches, kCntGateTests, kCntNodeVisits, kCntTableFlops,
       // The pooled traversal works on several entries of a ray at once and so enters subtrees that the hit it finds later would
       // have pruned.  These three count what a traversal that knew each ray's hit distance in advance visits (closest-hit traces:
       // a second pass bounded by the result; any-hit traces: what they did visit).
       kCntUsefulNodeTests, kCntUsefulTriTests, kCntUsefulNodeVisits, kCntCount };
constexpr int kCounterSlots = 256;
constexpr int kCounterStride = 16;          // 64-bit words from one set to the next
constexpr int kCounterWords = kCounterSlots * kCounterStride;

}  // namespace rt
