// C ABI of the renderer (see include/rt_api.h for the contract and the reference call sites each entry replaces).
#include "rt_api.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_build.h"
#include "rt_device_types.h"
#include "rt_kernels.h"
#include "rt_scene.h"

static_assert(RT_TILE == RT_TILE_PX, "public and internal tile sizes must agree");

namespace {

thread_local std::string g_lastError;

int fail(int code, const std::string& msg) { g_lastError = msg; return code; }

int failHip(hipError_t e, const char* what) {
    return fail(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define RT_HIP(call)                                         \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return failHip(e_, #call);     \
    } while (0)

template <typename T>
int uploadArray(const std::vector<T>& host, const T** devOut, std::vector<void*>& allocs, int64_t& bytes) {
    *devOut = nullptr;
    if (host.empty()) return RT_OK;
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, host.size() * sizeof(T)));
    allocs.push_back(p);
    RT_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    bytes += (int64_t)(host.size() * sizeof(T));
    *devOut = static_cast<const T*>(p);
    return RT_OK;
}

}  // namespace

struct rt_ctx {
    int device = 0;
    bool haveScene = false, haveState = false;
    bool chunksSequential = true;
    int dealRank0 = 1, dealOther = 1;   // how the tiles of a sharded frame are dealt to the ranks (rt_set_tile_deal)
    rt::DevScene scene{};
    rt::FrameParams frame{};
    rt_scene_info info{};
    std::vector<void*> sceneAllocs;
    // frame-owned scratch for the host-pointer entry points
    uint8_t* dFrame = nullptr; size_t frameCap = 0;
    int32_t* dIds = nullptr; size_t idsCap = 0;
    float* dT = nullptr; size_t tCap = 0;
    unsigned long long* dCounters = nullptr;
    unsigned long long* hCounters = nullptr;    // pinned: the counters of a frame come back with a copy that does not stage through pageable memory
    hipEvent_t evStart = nullptr, evStop = nullptr;
    // The blocking host entry points work on a stream of their own: operations on the default stream first wait for every
    // other stream of the process (0.29 ms per 1080p frame beside a PyTorch process's streams).
    hipStream_t hostStream = nullptr;
    // host copies of the small tables that rt_update_frame_state may rewrite
    rt::F4 sphere0Geom{}; bool hasSphere0 = false;
    rt::F4 plane0Mat0{}; bool hasPlane0 = false;
    float meshLo[3] = {0, 0, 0}, meshHi[3] = {0, 0, 0}; bool haveMeshBounds = false;   // bounds of every reachable triangle
    bool sphere0Dirty = false, plane0Dirty = false;   // the device tables still hold older values: the next launch patches them first
    int blockOrder = 2;
    int nodePoolEntries = 0;          // 0 = the kernel's compiled capacity
    int leadWaves = 2048;             // regular waves ahead of the split waves in the grid (RT_LEAD_WAVES; measured best of 0 .. all)
    int traversal = (int)(rt::kTravDirectStart | rt::kTravExpandSiblings | (8u << 8));   // kTrav* bits | expansion limit / 8 << 8 (RT_TRAV overrides)
    // Cost-ordered dispatch: per-tile cost of the previous frame and the order derived from it (device, fixed capacity
    // so that the stream entry points never allocate).  orderKey identifies the frame geometry the order belongs to.
    uint32_t* dTileCost = nullptr;    // two buffers of kMaxOrderedTiles, used alternately
    uint32_t* dTileOrder = nullptr;
    int costSlot = 0;
    long long orderKey = -1;
    bool costOrdering = true;
    // Block feedback (see rt::BlockFeedback): the heaviest 8x8 blocks of a frame are rendered by `split` waves in the next one.
    uint32_t* dBlockCost = nullptr;   // kMaxSplitBlocks
    uint32_t* dExtraMap = nullptr;    // kMaxExtraWaves
    uint32_t* dHeavyBits = nullptr;   // kMaxSplitBlocks / 32 + 8
    rt::SplitState* dSplitState = nullptr;
    int splitSlot = 0;                // slot of dSplitState the next selection writes
    int framesSinceRank = 0, rankedFrames = 0;   // see kRankPeriod
    int framesSinceSelection = 0;     // while nothing is split the selection only runs every kIdleSelectionPeriod-th frame
    int split = 1;                    // 0 disables splitting
    int splitRatio256 = 512;          // a block is split if it costs more than this share (in 1/256) of the evenly spread frame time
    int splitStartTenths = 16;        // splitting starts when the heaviest block exceeds this many tenths of the evenly packed frame
    uint32_t splitFloor = 4500;       // ... and at least this much (wave run time, 10 ns units)
    int residentWaves = 0;
    // Host-mapped word the selection kernel writes the number of split waves to.  Read without synchronising: a stale value
    // only decides whether a frame uses split waves at all (with split = 0 the regular waves render every block).
    volatile uint32_t* hSplitWaves = nullptr;
    uint32_t* dSplitWaves = nullptr;
    float* dPathCams = nullptr;          // RT_MAX_PATH_FRAMES x 16 floats: the cameras of a camera-path launch
    // Pinned staging copies of the same, used in turn (the asynchronous copy reads one after the call has returned, and it runs
    // only when the launch before it on that stream is done: with a single staging buffer every path call would wait for the
    // previous launch to finish).  pathCamsFree[i] is recorded after the copy out of slot i; a call waits for its slot's event.
    static constexpr int kPathStaging = 4;
    float* hPathCams = nullptr;
    hipEvent_t pathCamsFree[kPathStaging] = {nullptr, nullptr, nullptr, nullptr};
    unsigned pathCalls = 0;
    rt::StaticScratch* staticScratch = nullptr;   // this device's record of the never-allocated scratch (rt_render_device_static)
    // The stream whose launches last used the context's dispatch feedback and scene tables (see orderAfterOtherStreams).
    hipStream_t lastStream = nullptr; bool lastStreamValid = false;
    bool strictOctreeFlag = false;   // config.octree hides meshes, as the reference's unfinished branch does
    bool forceDeviceBuild = false;   // RT_DEVICE_BUILD=1: build every scene's octrees on the GPU
    // what rt_validate_scene_device needs besides the device arrays
    std::vector<uint32_t> mdlInfoHost;
    int64_t nodeFloat4s = 0, triFloat4s = 0;
};

namespace {

void freeScene(rt_ctx* c) {
    for (void* p : c->sceneAllocs) (void)hipFree(p);
    c->sceneAllocs.clear();
    c->scene = rt::DevScene{};
    c->haveScene = false;
}

constexpr int kRankPeriod = 8;          // tiles are ranked after every kRankPeriod-th frame ...
constexpr int kRankWarmup = 3;          // ... once the first kRankWarmup frames of a geometry have each been ranked
constexpr int kIdleSelectionPeriod = 32;
constexpr int kActiveSelectionPeriod = 2;
constexpr int kMaxOrderedTiles = 1 << 16;   // frames with more tiles per rank are rendered in plain tile order

int tilesAlong(int pixels) { return (pixels + RT_TILE - 1) / RT_TILE; }

rt::TileDeal dealOf(const rt_ctx* ctx, int world) { return rt::TileDeal{world, ctx->dealRank0, ctx->dealOther}; }

bool validDeal(int rank0Slots, int otherSlots) { return rank0Slots >= 1 && rank0Slots <= 64 && otherSlots >= 1 && otherSlots <= 64; }

// bytes of one rank's tile buffer under the context's deal
size_t tileBufferBytes(const rt_ctx* ctx, int width, int height, int world) {
    return rt_tile_buffer_bytes_dealt(width, height, world, ctx->dealRank0, ctx->dealOther);
}

// The dispatch feedback (tile costs, the tile order ranked from them, the block lists) and the patched scene tables are written
// by the launches of one call and read by those of the next.  Launches on one stream are ordered.  A call that arrives on
// another stream than the last user's must not start while that one's launches are queued or running: a render launch that
// reads a tile order while a ranking kernel rewrites it, or a ranking kernel that reads costs while a render launch raises
// them, sees a list that is no permutation, and tiles go unrendered.  The new stream cannot be made to wait for the old one
// (the caller may have destroyed it), so the host waits for the device.  A stream switch is rare - a context normally lives
// on one stream - and costs nothing otherwise.
int orderAfterOtherStreams(hipStream_t* last, bool* valid, hipStream_t stream) {
    if (*valid && *last != stream) RT_HIP(hipDeviceSynchronize());
    *last = stream;
    *valid = true;
    return RT_OK;
}

bool cameraWithinNoBoxRange(const rt_ctx* ctx, const float* camPos) {
    // Without a bounding-box gate Model::hit also accepts hits behind the ray origin at any distance (Model.cpp:318-356),
    // and the padding of the octree boxes covers the traversal's rounding error only up to about 3000 units between a ray
    // origin and a mesh.  Primary and reflection rays start within 999 units of the camera.
    float d2 = 0.0f;
    for (int a = 0; a < 3; a++) {
        const float c = camPos[a];
        const float d = c < ctx->meshLo[a] ? ctx->meshLo[a] - c : (c > ctx->meshHi[a] ? c - ctx->meshHi[a] : 0.0f);
        d2 += d * d;
    }
    return d2 <= 1500.0f * 1500.0f;      // false for NaN
}

int checkFrameArgs(rt_ctx* ctx, int width, int height) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) return fail(RT_ERR_INVALID, "frame size out of range");
    if (!ctx->haveScene) return fail(RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (!ctx->haveState) return fail(RT_ERR_NO_STATE, "rt_update_frame_state has not been called");
    if (ctx->frame.meshMode == rt::kMeshEightBoxes && !ctx->chunksSequential)
        return fail(RT_ERR_INVALID, "eightBoundingBoxes needs models whose chunk boxes list their triangles in file order");
    if (ctx->frame.meshMode == rt::kMeshNoBox && ctx->haveMeshBounds && !cameraWithinNoBoxRange(ctx, ctx->frame.camPos))
        return fail(RT_ERR_INVALID, "boundingBox = false: the camera must stay within 1500 units of the meshes");
    // every entry point launches on the context's own device, whatever device the calling thread had current
    RT_HIP(hipSetDevice(ctx->device));
    return RT_OK;
}

rt::FrameParams frameFor(const rt_ctx* ctx, int width, int height, int rank, int world, bool compact) {
    rt::FrameParams p = ctx->frame;
    p.width = width; p.height = height;
    const rt::TileDeal deal = dealOf(ctx, world);
    p.rank = rt::dealFirstSlot(deal, rank); p.world = rt::dealPeriod(deal); p.tileRun = rt::dealRun(deal, rank);
    p.tilesX = tilesAlong(width); p.tilesY = tilesAlong(height);
    p.ownedTiles = rt::dealOwned(deal, rank, p.tilesX * p.tilesY);
    p.compactOutput = compact ? 1 : 0;
    p.blockOrder = ctx->blockOrder;
    p.nodePoolEntries = ctx->nodePoolEntries;
    p.traversal = ctx->traversal;
    p.leadWaves = ctx->leadWaves;
    p.split = 0;
    return p;
}

// Launches the render kernel for `p` with cost-ordered dispatch: tiles are started in the order of their measured cost
// on the previous frame of the same geometry, and this frame's costs are ranked afterwards for the next one.
// Decides how the frame is dispatched: whether tiles are cost-ordered and whether an order from the previous frame of the
// same geometry exists.
struct DispatchPlan { bool ordered, haveOrder; long long key; rt::FrameParams p; };

DispatchPlan planDispatch(const rt_ctx* ctx, const rt::FrameParams& p) {
    DispatchPlan d;
    d.p = p;
    d.ordered = ctx->costOrdering && p.blockOrder == 2 && p.ownedTiles > 0 && p.ownedTiles <= kMaxOrderedTiles;
    // frame geometry the tile order belongs to: 16 bits each of width and height, 12 bits each of world and rank (wider
    // values only cost a needless restart of the feedback), and the tile count the order list was sized for
    d.key = ((long long)(p.width & 0xffff) << 47) | ((long long)(p.height & 0xffff) << 31) | ((long long)(p.world & 0xfff) << 19) |
            ((long long)(p.rank & 0xfff) << 7) | (long long)(p.tileRun & 0x7f);
    d.key ^= (long long)p.ownedTiles * 0x9E3779B1LL;
    d.haveOrder = d.ordered && ctx->orderKey == d.key;
    return d;
}

// Frame parameters as launchOrdered will launch them: blocks are split only on top of an existing tile order.
rt::FrameParams launchParams(const rt_ctx* ctx, const DispatchPlan& plan, bool* feedbackOut = nullptr) {
    rt::FrameParams launch = plan.p;
    const int blocks = plan.p.ownedTiles * (RT_TILE / 8) * (RT_TILE / 8);
    const bool feedback = plan.ordered && ctx->split > 0 && blocks <= rt::kMaxSplitBlocks;
    launch.split = (feedback && plan.haveOrder && ctx->hSplitWaves && *ctx->hSplitWaves != 0u) ? 1 : 0;
    if (feedbackOut) *feedbackOut = feedback;
    return launch;
}

int launchOrdered(rt_ctx* ctx, const rt::FrameParams& frame, uint8_t* out, int32_t* ids, float* tbuf, unsigned long long* counters,
                  bool instrumented, hipStream_t stream, unsigned long long* waveClocks = nullptr) {
    const DispatchPlan plan = planDispatch(ctx, frame);
    const rt::FrameParams& p = plan.p;
    if (int rc = orderAfterOtherStreams(&ctx->lastStream, &ctx->lastStreamValid, stream)) return rc;
    if (ctx->sphere0Dirty || ctx->plane0Dirty) {
        RT_HIP(rt::launchPatchTables(ctx->sphere0Dirty ? const_cast<rt::F4*>(ctx->scene.sphGeom) : nullptr, ctx->sphere0Geom,
                                     ctx->plane0Dirty ? const_cast<rt::F4*>(ctx->scene.plnMat) : nullptr, ctx->plane0Mat0, stream));
        ctx->sphere0Dirty = ctx->plane0Dirty = false;
    }
    if (!plan.ordered) {
        RT_HIP(rt::launchRender(ctx->scene, p, out, ids, tbuf, counters, instrumented, stream, waveClocks));
        return RT_OK;
    }
    // Two cost buffers: this frame accumulates into `cur`; the ranking pass that follows reads it and clears `nxt`, which the
    // next frame accumulates into.  A change of frame geometry starts over from cleared buffers and plain tile order.
    const int blocks = p.ownedTiles * (RT_TILE / 8) * (RT_TILE / 8);
    bool feedback = false;
    const rt::FrameParams launch = launchParams(ctx, plan, &feedback);
    if (!plan.haveOrder) {
        RT_HIP(hipMemsetAsync(ctx->dTileCost, 0, sizeof(uint32_t) * kMaxOrderedTiles * 2, stream));
        if (feedback) {
            RT_HIP(hipMemsetAsync(ctx->dBlockCost, 0, sizeof(uint32_t) * (size_t)blocks, stream));
            RT_HIP(hipMemsetAsync(ctx->dSplitState, 0, sizeof(rt::SplitState), stream));
        }
    }
    uint32_t* cur = ctx->dTileCost + (size_t)ctx->costSlot * kMaxOrderedTiles;
    uint32_t* nxt = ctx->dTileCost + (size_t)(ctx->costSlot ^ 1) * kMaxOrderedTiles;
    // While the last selection split nothing (host hint, possibly a frame or two old), the block costs are looked at only now
    // and then: a frame that does not split saves two small launches, and a frame that neither splits nor is about to be looked
    // at runs without the feedback altogether (the per-wave cost records and the split-list checks are 1.2 % of a 1080p frame).
    // The frame before a selection records the costs; the frame after it carries the feedback too, because that is where the
    // kernel reports the new selection to the host.
    static const int activePeriod = std::getenv("RT_SELECT_PERIOD") ? std::atoi(std::getenv("RT_SELECT_PERIOD")) : kActiveSelectionPeriod;
    static const int idlePeriod = std::getenv("RT_IDLE_SELECT_PERIOD") ? std::max(1, std::atoi(std::getenv("RT_IDLE_SELECT_PERIOD"))) : kIdleSelectionPeriod;
    const bool idle = !launch.split && ctx->hSplitWaves && *ctx->hSplitWaves == 0u;
    // ... and while blocks are split, every second frame: the choice changes slowly, the costs keep their maxima, and the two
    // launches are 7 % of the time a rank needs for an eighth of the 1080p frame (RT_SELECT_PERIOD overrides).
    const bool selectNow = feedback && (!plan.haveOrder || ctx->framesSinceSelection + 1 >= (idle ? idlePeriod : activePeriod));
    const bool record = feedback && (!idle || selectNow || ctx->framesSinceSelection == 0);
    rt::BlockFeedback fb{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (record) {
        fb.blockCost = ctx->dBlockCost;
        // the selection after the previous frame wrote slot splitSlot ^ 1
        fb.extraCount = &ctx->dSplitState->used[ctx->splitSlot ^ 1];
        fb.hint = ctx->dSplitWaves;
        fb.frameStart = &ctx->dSplitState->frameStart;
        if (launch.split) { fb.extraMap = ctx->dExtraMap; fb.heavyBits = ctx->dHeavyBits; }
    }
    // with launch.split the grid starts with the split waves of the heaviest blocks (their records follow the regular ones)
    RT_HIP(rt::launchRender(ctx->scene, launch, out, ids, tbuf, counters, instrumented, stream, waveClocks,
                            plan.haveOrder ? ctx->dTileOrder : nullptr, cur, fb));
    // The tiles are ranked after each of the first frames of a geometry, then after every eighth (RT_RANK_PERIOD overrides): the
    // costs keep accumulating their maxima in between, the order changes slowly, and the ranking launch with the gap around it is
    // 3 % of a 1080p frame that nothing else would wait for.
    static const int rankPeriod = std::getenv("RT_RANK_PERIOD") ? std::max(1, std::atoi(std::getenv("RT_RANK_PERIOD"))) : kRankPeriod;
    const bool rank = !plan.haveOrder || ctx->rankedFrames < kRankWarmup || ++ctx->framesSinceRank >= rankPeriod;
    if (!plan.haveOrder) ctx->rankedFrames = 0;
    if (rank) {
        RT_HIP(rt::launchRankTiles(cur, ctx->dTileOrder, nxt, p.ownedTiles, stream));
        ctx->framesSinceRank = 0;
        if (ctx->rankedFrames < kRankWarmup) ctx->rankedFrames++;
    }
    if (feedback && !selectNow) {
        ctx->framesSinceSelection++;
    } else if (feedback) {
        ctx->framesSinceSelection = 0;
        RT_HIP(rt::launchSelectHeavy(ctx->dBlockCost, ctx->dExtraMap, ctx->dHeavyBits, blocks, ctx->dSplitState, ctx->splitSlot, ctx->residentWaves,
                                     ctx->splitRatio256, ctx->splitFloor, ctx->splitStartTenths, stream));
        ctx->splitSlot ^= 1;
    }
    if (rank) ctx->costSlot ^= 1;
    ctx->orderKey = plan.key;
    return RT_OK;
}

template <typename T>
int ensureCapacity(T** buf, size_t* cap, size_t need) {
    if (*cap >= need) return RT_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    void* p = nullptr;
    RT_HIP(hipMalloc(&p, need * sizeof(T)));
    *buf = static_cast<T*>(p);
    *cap = need;
    return RT_OK;
}

int renderToHost(rt_ctx* ctx, int width, int height, uint8_t* rgbOut, int32_t* idsOut, float* tOut, bool instrumented,
                 rt_frame_stats* stats) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t pixels = (size_t)width * (size_t)height;
    rc = ensureCapacity(&ctx->dFrame, &ctx->frameCap, pixels * 3);
    if (rc) return rc;
    if (idsOut) { rc = ensureCapacity(&ctx->dIds, &ctx->idsCap, pixels * 2); if (rc) return rc; }
    if (tOut) { rc = ensureCapacity(&ctx->dT, &ctx->tCap, pixels); if (rc) return rc; }

    rt::FrameParams p = frameFor(ctx, width, height, 0, 1, false);
    if (instrumented) p.flags |= rt::kCfgCountUseful;
    const hipStream_t hs = ctx->hostStream;
    RT_HIP(hipMemsetAsync(ctx->dCounters, 0, sizeof(unsigned long long) * rt::kCounterWords, hs));
    RT_HIP(hipEventRecord(ctx->evStart, hs));
    rc = launchOrdered(ctx, p, ctx->dFrame, idsOut ? ctx->dIds : nullptr, tOut ? ctx->dT : nullptr, ctx->dCounters, instrumented, hs);
    if (rc) return rc;
    RT_HIP(hipEventRecord(ctx->evStop, hs));
    if (rgbOut) RT_HIP(hipMemcpyAsync(rgbOut, ctx->dFrame, pixels * 3, hipMemcpyDeviceToHost, hs));
    if (idsOut) RT_HIP(hipMemcpyAsync(idsOut, ctx->dIds, pixels * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, hs));
    if (tOut) RT_HIP(hipMemcpyAsync(tOut, ctx->dT, pixels * sizeof(float), hipMemcpyDeviceToHost, hs));
    RT_HIP(hipMemcpyAsync(ctx->hCounters, ctx->dCounters, sizeof(unsigned long long) * rt::kCounterWords, hipMemcpyDeviceToHost, hs));
    RT_HIP(hipStreamSynchronize(hs));
    unsigned long long counters[rt::kCntCount] = {};
    for (int slot = 0; slot < rt::kCounterSlots; slot++)
        for (int c = 0; c < rt::kCntCount; c++) counters[c] += ctx->hCounters[slot * rt::kCounterStride + c];
    if (stats) {
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop));
        stats->gpu_ms = ms;
        stats->rays = counters[rt::kCntRays];
        stats->node_tests = counters[rt::kCntNodeTests];
        stats->tri_tests = counters[rt::kCntTriTests];
        stats->attr_fetches = counters[rt::kCntAttrFetches];
        stats->gate_tests = counters[rt::kCntGateTests];
        stats->node_visits = counters[rt::kCntNodeVisits];
        stats->table_flops = counters[rt::kCntTableFlops];
        stats->useful_node_tests = counters[rt::kCntUsefulNodeTests];
        stats->useful_tri_tests = counters[rt::kCntUsefulTriTests];
        stats->useful_node_visits = counters[rt::kCntUsefulNodeVisits];
        stats->split_waves = ctx->hSplitWaves ? *ctx->hSplitWaves : 0u;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_lastError.c_str(); }

int rt_create(int device_count, const int* devices, rt_ctx** out) {
    if (!out) return fail(RT_ERR_INVALID, "null output pointer");
    *out = nullptr;
    if (device_count != 1 || !devices) return fail(RT_ERR_INVALID, "device_count must be 1 (one context per GPU)");
    int n = 0;
    RT_HIP(hipGetDeviceCount(&n));
    if (devices[0] < 0 || devices[0] >= n) return fail(RT_ERR_INVALID, "no such HIP device");
    RT_HIP(hipSetDevice(devices[0]));
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return fail(RT_ERR_INVALID, "out of host memory");
    c->device = devices[0];
    if (const char* e = std::getenv("RT_BLOCK_ORDER")) c->blockOrder = std::atoi(e);   // experiment switches
    if (const char* e = std::getenv("RT_COST_ORDER")) c->costOrdering = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_NODE_POOL")) c->nodePoolEntries = std::atoi(e) & 0xffff;    // tests: forces the traversal's low-room path
    if (const char* e = std::getenv("RT_TRI_POOL")) c->nodePoolEntries |= (std::atoi(e) & 0xffff) << 16;  // the same for the pool of triangle work
    if (const char* e = std::getenv("RT_TRAV")) c->traversal = std::atoi(e) & 0xffff;   // the kernel caps the expansion limit at its pool size
    if (const char* e = std::getenv("RT_LEAD_WAVES")) { const int v = std::atoi(e); if (v >= 0) c->leadWaves = (v + 7) / 8 * 8; }
    if (const char* e = std::getenv("RT_DEVICE_BUILD")) c->forceDeviceBuild = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_SPLIT")) c->split = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("RT_SPLIT_RATIO")) { const int v = std::atoi(e); if (v >= 0 && v <= 65536) c->splitRatio256 = v; }
    {   // once per device, with that device current: the address of this GPU's copy of the static scratch array
        const hipError_t se = rt::staticScratch(c->device, &c->staticScratch);
        if (se != hipSuccess) { const int rc = failHip(se, "hipGetSymbolAddress(static scratch)"); rt_destroy(c); return rc; }
    }
    c->residentWaves = rt::residentRenderWaves();
    if (c->residentWaves <= 0) c->split = 0;
    if (c->split) {
        hipError_t se = hipSuccess;
        void* hp = nullptr;
        if (se == hipSuccess && !std::getenv("RT_NO_HINT")) se = hipHostMalloc(&hp, 64, hipHostMallocMapped);
        if (se == hipSuccess && hp) {
            c->hSplitWaves = static_cast<volatile uint32_t*>(hp);
            *c->hSplitWaves = 0u;
            void* dp = nullptr;
            se = hipHostGetDevicePointer(&dp, hp, 0);
            c->dSplitWaves = static_cast<uint32_t*>(dp);
        }
        if (se != hipSuccess) { const int rc = failHip(se, "host-visible hint for split waves"); rt_destroy(c); return rc; }
    }
    if (const char* e = std::getenv("RT_SPLIT_START")) { const int v = std::atoi(e); if (v >= 4 && v <= 1000) c->splitStartTenths = v; }
    if (const char* e = std::getenv("RT_SPLIT_FLOOR")) { const int v = std::atoi(e); if (v >= 0) c->splitFloor = (uint32_t)v; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sizeof(unsigned long long) * rt::kCounterWords);
    if (e != hipSuccess) { const int rc = failHip(e, "hipMalloc(counters)"); rt_destroy(c); return rc; }
    c->dCounters = static_cast<unsigned long long*>(p);
    e = hipHostMalloc(&p, sizeof(unsigned long long) * rt::kCounterWords, hipHostMallocDefault);
    if (e != hipSuccess) { const int rc = failHip(e, "hipHostMalloc(counters)"); rt_destroy(c); return rc; }
    c->hCounters = static_cast<unsigned long long*>(p);
    void* q = nullptr;
    e = hipMalloc(&q, sizeof(uint32_t) * kMaxOrderedTiles * 2);
    if (e == hipSuccess) { c->dTileCost = static_cast<uint32_t*>(q); e = hipMemset(q, 0, sizeof(uint32_t) * kMaxOrderedTiles * 2); }
    if (e == hipSuccess) { e = hipMalloc(&q, sizeof(uint32_t) * kMaxOrderedTiles); if (e == hipSuccess) c->dTileOrder = static_cast<uint32_t*>(q); }
    if (e == hipSuccess) { e = hipMalloc(&q, sizeof(uint32_t) * rt::kMaxSplitBlocks); if (e == hipSuccess) c->dBlockCost = static_cast<uint32_t*>(q); }
    if (e == hipSuccess) { e = hipMalloc(&q, sizeof(uint32_t) * rt::kMaxExtraWaves); if (e == hipSuccess) c->dExtraMap = static_cast<uint32_t*>(q); }
    if (e == hipSuccess) { e = hipMalloc(&q, sizeof(uint32_t) * (rt::kMaxSplitBlocks / 32 + 8)); if (e == hipSuccess) c->dHeavyBits = static_cast<uint32_t*>(q); }
    if (e == hipSuccess) { e = hipMalloc(&q, sizeof(rt::SplitState)); if (e == hipSuccess) { c->dSplitState = static_cast<rt::SplitState*>(q); e = hipMemset(q, 0, sizeof(rt::SplitState)); } }
    if (e != hipSuccess) { rt_destroy(c); return failHip(e, "hipMalloc(tile order)"); }
    e = hipMalloc(&q, sizeof(float) * 16 * RT_MAX_PATH_FRAMES);
    if (e == hipSuccess) { c->dPathCams = static_cast<float*>(q); e = hipHostMalloc(&q, sizeof(float) * 16 * RT_MAX_PATH_FRAMES * rt_ctx::kPathStaging, hipHostMallocDefault); }
    if (e == hipSuccess) c->hPathCams = static_cast<float*>(q);
    for (int i = 0; i < rt_ctx::kPathStaging && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->pathCamsFree[i], hipEventDisableTiming);
    if (e != hipSuccess) { rt_destroy(c); return failHip(e, "camera-path buffers"); }
    e = hipStreamCreateWithFlags(&c->hostStream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->evStart);
    if (e == hipSuccess) e = hipEventCreate(&c->evStop);
    if (e != hipSuccess) { rt_destroy(c); return failHip(e, "hipStreamCreate / hipEventCreate"); }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    freeScene(ctx);
    if (ctx->dFrame) (void)hipFree(ctx->dFrame);
    if (ctx->dIds) (void)hipFree(ctx->dIds);
    if (ctx->dT) (void)hipFree(ctx->dT);
    if (ctx->dCounters) (void)hipFree(ctx->dCounters);
    if (ctx->hCounters) (void)hipHostFree(ctx->hCounters);
    if (ctx->dTileCost) (void)hipFree(ctx->dTileCost);
    if (ctx->dTileOrder) (void)hipFree(ctx->dTileOrder);
    if (ctx->dBlockCost) (void)hipFree(ctx->dBlockCost);
    if (ctx->dExtraMap) (void)hipFree(ctx->dExtraMap);
    if (ctx->dHeavyBits) (void)hipFree(ctx->dHeavyBits);
    if (ctx->dSplitState) (void)hipFree(ctx->dSplitState);
    if (ctx->hSplitWaves) (void)hipHostFree(const_cast<uint32_t*>(ctx->hSplitWaves));
    if (ctx->dPathCams) (void)hipFree(ctx->dPathCams);
    if (ctx->hPathCams) (void)hipHostFree(ctx->hPathCams);
    for (hipEvent_t ev : ctx->pathCamsFree) if (ev) (void)hipEventDestroy(ev);
    if (ctx->hostStream) (void)hipStreamDestroy(ctx->hostStream);
    if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
    if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
    delete ctx;
}

int rt_upload_scene(rt_ctx* ctx, const rt_scene_desc* scene) {
    if (!ctx || !scene) return fail(RT_ERR_INVALID, "null argument");
    const bool onDevice = scene->buildOnDevice != 0 || ctx->forceDeviceBuild;
    rt::FlatScene flat;
    std::string err;
    if (!rt::flattenScene(*scene, flat, err, onDevice)) return fail(RT_ERR_INVALID, err);
    // octrees that came from the caller are checked against the model they claim to belong to before the GPU walks them
    if (flat.prebuiltOctrees > 0 && !rt::validateFlatScene(*scene, flat, err)) return fail(RT_ERR_INVALID, "supplied octree: " + err);

    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());
    freeScene(ctx);
    rt::DevScene d{};
    int64_t bytes = 0;
    int rc = 0;
    auto& A = ctx->sceneAllocs;
    if ((rc = uploadArray(flat.sphGeom, &d.sphGeom, A, bytes)) || (rc = uploadArray(flat.sphMat, &d.sphMat, A, bytes)) ||
        (rc = uploadArray(flat.plnP, &d.plnP, A, bytes)) || (rc = uploadArray(flat.plnN, &d.plnN, A, bytes)) ||
        (rc = uploadArray(flat.plnMat, &d.plnMat, A, bytes)) || (rc = uploadArray(flat.plnNormals, &d.plnNormals, A, bytes)) || (rc = uploadArray(flat.boxMin, &d.boxMin, A, bytes)) ||
        (rc = uploadArray(flat.boxMax, &d.boxMax, A, bytes)) || (rc = uploadArray(flat.boxMat, &d.boxMat, A, bytes)) ||
        (rc = uploadArray(flat.mdlGateMin, &d.mdlGateMin, A, bytes)) || (rc = uploadArray(flat.mdlGateMax, &d.mdlGateMax, A, bytes)) ||
        (rc = uploadArray(flat.mdlTightMin, &d.mdlTightMin, A, bytes)) || (rc = uploadArray(flat.mdlTightMax, &d.mdlTightMax, A, bytes)) ||
        (rc = uploadArray(flat.mdlMat, &d.mdlMat, A, bytes)) || (rc = uploadArray(flat.mdlInfo, &d.mdlInfo, A, bytes)) ||
        (rc = uploadArray(flat.mdlChunkMin, &d.mdlChunkMin, A, bytes)) || (rc = uploadArray(flat.mdlChunkMax, &d.mdlChunkMax, A, bytes))) {
        freeScene(ctx);
        return rc;
    }
    if (onDevice && !flat.deviceModels.empty()) {
        rt::DeviceMeshes built;
        hipError_t e = rt::buildMeshesOnDevice(*scene, flat, built, err);
        if (e != hipSuccess) { freeScene(ctx); return fail(e == hipErrorInvalidValue ? RT_ERR_INVALID : RT_ERR_DEVICE, err); }
        A.push_back(built.nodes); A.push_back(built.tris); A.push_back(built.attrs);
        d.nodes = built.nodes; d.tris = built.tris; d.attrs = built.attrs;
        flat.nodeCount = built.nodeCount; flat.leafCount = built.leafCount;
        flat.maxDepth = flat.forestDepth + built.depth;
        ctx->nodeFloat4s = built.nodeCount * rt::kNodeFloat4s;
        ctx->triFloat4s = built.triangleRecords * 3;
        bytes += (ctx->nodeFloat4s + ctx->triFloat4s + flat.triangleCount * rt::kAttrFloat4s) * (int64_t)sizeof(rt::F4);
        if (flat.maxDepth >= rt::kMaxOctreeDepth) { freeScene(ctx); return fail(RT_ERR_INVALID, "forest plus octree nesting exceeds the depth the traversal reserves pool entries for"); }
    } else {
        if ((rc = uploadArray(flat.nodes, &d.nodes, A, bytes)) || (rc = uploadArray(flat.tris, &d.tris, A, bytes)) ||
            (rc = uploadArray(flat.attrs, &d.attrs, A, bytes))) {
            freeScene(ctx);
            return rc;
        }
        ctx->nodeFloat4s = (int64_t)flat.nodes.size();
        ctx->triFloat4s = (int64_t)flat.tris.size();
    }
    // the traversal keeps a node index in 29 bits of a pool entry (512 M nodes of 128 bytes would be 64 GB)
    if (ctx->nodeFloat4s / rt::kNodeFloat4s >= ((int64_t)1 << 29)) { freeScene(ctx); return fail(RT_ERR_CAPACITY, "more than 2^29 octree nodes"); }
    ctx->mdlInfoHost = flat.mdlInfo;
    d.nSph = scene->sphereCount; d.nPln = scene->planeCount; d.nBox = scene->AABBCount; d.nMdl = scene->modelCount;
    d.forestRoot = flat.forestRoot;
    std::memcpy(d.lightPos, flat.lightPos, sizeof d.lightPos);
    std::memcpy(d.lightCol, flat.lightCol, sizeof d.lightCol);
    ctx->scene = d;
    ctx->haveScene = true;
    ctx->chunksSequential = flat.chunksSequential;
    ctx->hasSphere0 = !flat.sphGeom.empty();
    if (ctx->hasSphere0) ctx->sphere0Geom = flat.sphGeom[0];
    ctx->sphere0Dirty = ctx->plane0Dirty = false;
    ctx->haveMeshBounds = false;
    for (size_t i = 0; i < flat.mdlTightMin.size(); i++) {
        if (flat.mdlInfo[4 * i] == rt::kNoRoot) continue;
        const float lo[3] = {flat.mdlTightMin[i].x, flat.mdlTightMin[i].y, flat.mdlTightMin[i].z};
        const float hi[3] = {flat.mdlTightMax[i].x, flat.mdlTightMax[i].y, flat.mdlTightMax[i].z};
        for (int a = 0; a < 3; a++) {
            ctx->meshLo[a] = ctx->haveMeshBounds ? std::min(ctx->meshLo[a], lo[a]) : lo[a];
            ctx->meshHi[a] = ctx->haveMeshBounds ? std::max(ctx->meshHi[a], hi[a]) : hi[a];
        }
        ctx->haveMeshBounds = true;
    }
    ctx->hasPlane0 = !flat.plnMat.empty();
    if (ctx->hasPlane0) ctx->plane0Mat0 = flat.plnMat[0];

    rt_scene_info& in = ctx->info;
    in.sphereCount = d.nSph; in.planeCount = d.nPln; in.AABBCount = d.nBox; in.modelCount = d.nMdl;
    in.triangleCount = flat.triangleCount; in.nodeCount = flat.nodeCount; in.leafCount = flat.leafCount;
    in.maxDepth = flat.maxDepth; in.deviceBytes = bytes;
    return RT_OK;
}

int rt_update_objects(rt_ctx* ctx, const rt_scene_desc* scene) {
    if (!ctx || !scene) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->haveScene) return fail(RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    rt_scene_desc only = *scene;           // the tables of spheres, planes and boxes and the light; the uploaded meshes stay
    only.models = nullptr; only.modelCount = 0;
    rt::FlatScene flat;
    std::string err;
    if (!rt::flattenScene(only, flat, err, false)) return fail(RT_ERR_INVALID, err);

    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipDeviceSynchronize());        // launches that still read the old tables
    rt::DevScene d = ctx->scene;
    std::vector<void*> fresh;
    int64_t bytes = 0;
    int rc = 0;
    if ((rc = uploadArray(flat.sphGeom, &d.sphGeom, fresh, bytes)) || (rc = uploadArray(flat.sphMat, &d.sphMat, fresh, bytes)) ||
        (rc = uploadArray(flat.plnP, &d.plnP, fresh, bytes)) || (rc = uploadArray(flat.plnN, &d.plnN, fresh, bytes)) ||
        (rc = uploadArray(flat.plnMat, &d.plnMat, fresh, bytes)) || (rc = uploadArray(flat.plnNormals, &d.plnNormals, fresh, bytes)) ||
        (rc = uploadArray(flat.boxMin, &d.boxMin, fresh, bytes)) || (rc = uploadArray(flat.boxMax, &d.boxMax, fresh, bytes)) ||
        (rc = uploadArray(flat.boxMat, &d.boxMat, fresh, bytes))) {
        for (void* p : fresh) (void)hipFree(p);       // the context keeps the scene it had
        return rc;
    }
    const void* old[9] = {ctx->scene.sphGeom, ctx->scene.sphMat, ctx->scene.plnP, ctx->scene.plnN, ctx->scene.plnMat, ctx->scene.plnNormals,
                          ctx->scene.boxMin, ctx->scene.boxMax, ctx->scene.boxMat};
    int64_t oldBytes = ((int64_t)ctx->scene.nSph * 4 + (int64_t)ctx->scene.nPln * (5 + 5) + (int64_t)ctx->scene.nBox * 5) * (int64_t)sizeof(rt::F4);
    auto& A = ctx->sceneAllocs;
    for (const void* o : old) {
        if (!o) continue;
        for (size_t i = 0; i < A.size(); i++) if (A[i] == o) { (void)hipFree(A[i]); A.erase(A.begin() + (long)i); break; }
    }
    A.insert(A.end(), fresh.begin(), fresh.end());
    d.nSph = scene->sphereCount; d.nPln = scene->planeCount; d.nBox = scene->AABBCount;
    std::memcpy(d.lightPos, flat.lightPos, sizeof d.lightPos);
    std::memcpy(d.lightCol, flat.lightCol, sizeof d.lightCol);
    ctx->scene = d;
    ctx->hasSphere0 = !flat.sphGeom.empty();
    if (ctx->hasSphere0) ctx->sphere0Geom = flat.sphGeom[0];
    ctx->hasPlane0 = !flat.plnMat.empty();
    if (ctx->hasPlane0) ctx->plane0Mat0 = flat.plnMat[0];
    ctx->sphere0Dirty = ctx->plane0Dirty = false;
    rt_scene_info& in = ctx->info;
    in.sphereCount = d.nSph; in.planeCount = d.nPln; in.AABBCount = d.nBox;
    in.deviceBytes += bytes - oldBytes;
    return RT_OK;
}

int rt_validate_scene_device(rt_ctx* ctx, const rt_scene_desc* scene) {
    if (!ctx || !scene) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->haveScene) return fail(RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if ((size_t)scene->modelCount * 4 != ctx->mdlInfoHost.size()) return fail(RT_ERR_INVALID, "not the scene this context was uploaded from");
    RT_HIP(hipSetDevice(ctx->device));
    rt::FlatScene flat;
    flat.mdlInfo = ctx->mdlInfoHost;
    flat.forestRoot = ctx->scene.forestRoot;
    flat.nodes.resize((size_t)ctx->nodeFloat4s);
    flat.tris.resize((size_t)ctx->triFloat4s);
    if (ctx->nodeFloat4s) RT_HIP(hipMemcpy(flat.nodes.data(), ctx->scene.nodes, flat.nodes.size() * sizeof(rt::F4), hipMemcpyDeviceToHost));
    if (ctx->triFloat4s) RT_HIP(hipMemcpy(flat.tris.data(), ctx->scene.tris, flat.tris.size() * sizeof(rt::F4), hipMemcpyDeviceToHost));
    std::string err;
    if (!rt::validateFlatScene(*scene, flat, err)) return fail(RT_ERR_INVALID, err);
    return RT_OK;
}

int rt_update_frame_state(rt_ctx* ctx, const rt_camera* cam, const rt_config* cfg, const rt_dynamic* dyn) {
    if (!ctx || !cam || !cfg) return fail(RT_ERR_INVALID, "null argument");
    if (cfg->softShadowNum < 0 || cfg->softShadowNum > 4096) return fail(RT_ERR_INVALID, "softShadowNum out of range");
    rt::FrameParams& p = ctx->frame;
    const float* src[4] = {&cam->position.x, &cam->up.x, &cam->right.x, &cam->direction.x};
    float* dst[4] = {p.camPos, p.camUp, p.camRight, p.camDir};
    for (int i = 0; i < 4; i++) std::memcpy(dst[i], src[i], 3 * sizeof(float));
    p.fovFactor = cam->fovFactor;
    {   // kernels.cu:48,56: v = normalise(cameraPosition) = position / sqrt((x x + y y) + z z), IEEE float without contraction
        const float x = p.camPos[0], y = p.camPos[1], z = p.camPos[2];
        const float len = std::sqrt(x * x + y * y + z * z);
        p.eyeDir[0] = x / len; p.eyeDir[1] = y / len; p.eyeDir[2] = z / len;
    }
    p.shadowBias = cfg->shadowBias; p.shadowIntensity = cfg->shadowIntensity;
    p.sphereStrength = cfg->sphereReflectionStrength; p.planeStrength = cfg->planeReflectionStrength;
    p.boxStrength = cfg->AABBReflectionStrength; p.modelStrength = cfg->modelReflectionStrength;
    p.background[0] = cfg->backgroundCol.x; p.background[1] = cfg->backgroundCol.y; p.background[2] = cfg->backgroundCol.z;
    p.softRadius = (float)cfg->softShadowRadius;
    p.softNum = cfg->softShadowNum; p.maxDepth = cfg->maxDepth;
    uint32_t f = 0;
    if (cfg->renderHardShadows) f |= rt::kCfgHardShadows;
    if (cfg->renderSoftShadows) f |= rt::kCfgSoftShadows;
    if (cfg->areaLightSpecularEffect) f |= rt::kCfgAreaSpecular;
    if (cfg->reflections) f |= rt::kCfgReflections;
    if (cfg->reflectPlanes) f |= rt::kCfgReflectPlanes;
    if (cfg->renderAABBs) f |= rt::kCfgRenderBoxes;
    if (cfg->renderModels) f |= rt::kCfgRenderModels;
    if (cfg->ambientLighting) f |= rt::kCfgAmbient;
    if (cfg->diffuseLighting) f |= rt::kCfgDiffuse;
    if (cfg->specularLighting) f |= rt::kCfgSpecular;
    if (cfg->antiAliasing) f |= rt::kCfgAntiAlias;
    if (cfg->cullBackTriangles) f |= rt::kCfgCull;
    // Mesh modes, in Model::hit's own priority order (Model.cpp:161-357).  `octree` is honoured as "whole-model box gate
    // with the octree as accelerator": the reference's octree branch is an unfinished stub that hides every mesh.
    if (cfg->octree) p.meshMode = rt::kMeshWholeBox;
    else if (cfg->eightBoundingBoxes) p.meshMode = rt::kMeshEightBoxes;
    else if (cfg->boundingBox) p.meshMode = rt::kMeshWholeBox;
    else p.meshMode = rt::kMeshNoBox;
    if (!(cfg->octree && ctx->strictOctreeFlag)) f |= rt::kCfgModelsVisible;
    p.flags = f;

    // The dynamic block only rewrites host copies here.  The light travels by value with every launch (DevScene::lightPos);
    // sphere 0 and plane 0's ambient colour are written into the device tables by a one-thread kernel that the next render
    // call enqueues on its own stream ahead of the frame, so a frame that is still running keeps the values it was issued with.
    if (dyn && !ctx->haveScene) return fail(RT_ERR_NO_SCENE, "rt_update_frame_state: a dynamic block needs an uploaded scene");
    if (dyn) {
        if (dyn->setLight) {
            ctx->scene.lightPos[0] = dyn->lightPosition.x; ctx->scene.lightPos[1] = dyn->lightPosition.y; ctx->scene.lightPos[2] = dyn->lightPosition.z;
        }
        if (dyn->setSphere0 && ctx->hasSphere0) {
            ctx->sphere0Geom = rt::F4{dyn->sphere0Position.x, dyn->sphere0Position.y, dyn->sphere0Position.z, dyn->sphere0Radius};
            ctx->sphere0Dirty = true;
        }
        if (dyn->setPlane0 && ctx->hasPlane0) {
            ctx->plane0Mat0.x = dyn->plane0AmbientColour.x; ctx->plane0Mat0.y = dyn->plane0AmbientColour.y; ctx->plane0Mat0.z = dyn->plane0AmbientColour.z;
            ctx->plane0Dirty = true;
        }
    }
    ctx->haveState = true;
    return RT_OK;
}

int rt_set_strict_octree_flag(rt_ctx* ctx, int strict) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    ctx->strictOctreeFlag = strict != 0;
    return RT_OK;
}

int rt_render_frame(rt_ctx* ctx, int width, int height, uint8_t* rgb_out, rt_frame_stats* stats) {
    if (!rgb_out) return fail(RT_ERR_INVALID, "null output buffer");
    return renderToHost(ctx, width, height, rgb_out, nullptr, nullptr, false, stats);
}

int rt_render_frame_ids(rt_ctx* ctx, int width, int height, uint8_t* rgb_out, int32_t* ids_out, float* t_out, rt_frame_stats* stats) {
    return renderToHost(ctx, width, height, rgb_out, ids_out, t_out, true, stats);
}

int rt_count_rays(rt_ctx* ctx, int width, int height, rt_frame_stats* stats) {
    if (!stats) return fail(RT_ERR_INVALID, "null stats");
    return renderToHost(ctx, width, height, nullptr, nullptr, nullptr, true, stats);
}

int rt_profile_waves(rt_ctx* ctx, int width, int height, uint64_t* clocks_out, size_t capacity_pairs, size_t* waves_out) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (!clocks_out || !waves_out) return fail(RT_ERR_INVALID, "null argument");
    RT_HIP(hipSetDevice(ctx->device));
    rt::FrameParams p = frameFor(ctx, width, height, 0, 1, false);
    // regular waves first, then room for the split waves whether or not this frame ends up using them
    const size_t waves = (size_t)rt::launchGridBlocks(p) + (size_t)rt::kMaxExtraWaves;
    *waves_out = waves;
    if (capacity_pairs < waves) return fail(RT_ERR_CAPACITY, "clock buffer too small");
    rc = ensureCapacity(&ctx->dFrame, &ctx->frameCap, (size_t)width * (size_t)height * 3);
    if (rc) return rc;
    void* d = nullptr;
    RT_HIP(hipMalloc(&d, waves * RT_WAVE_RECORD_WORDS * sizeof(uint64_t)));
    hipError_t e = hipMemset(d, 0, waves * RT_WAVE_RECORD_WORDS * sizeof(uint64_t));
    if (e != hipSuccess) { (void)hipFree(d); return failHip(e, "rt_profile_waves"); }
    rc = launchOrdered(ctx, p, ctx->dFrame, nullptr, nullptr, nullptr, true, nullptr, static_cast<unsigned long long*>(d));
    if (rc) { (void)hipFree(d); return rc; }
    if (e == hipSuccess) e = hipMemcpy(clocks_out, d, waves * RT_WAVE_RECORD_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return failHip(e, "rt_profile_waves");
    return RT_OK;
}

int rt_render_device_static(rt_ctx* ctx, int width, int height, void* d_rgb, void* hip_stream) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (!d_rgb) return fail(RT_ERR_INVALID, "null device buffer");
    if (!ctx->staticScratch) return fail(RT_ERR_DEVICE, "the static scratch array was not found on this device");
    rt::FrameParams p = frameFor(ctx, width, height, 0, 1, false);
    if (p.ownedTiles > rt::kStaticTiles)
        return fail(RT_ERR_CAPACITY, "rt_render_device_static: frame has " + std::to_string(p.ownedTiles) + " tiles of " + std::to_string(RT_TILE) +
                                     " x " + std::to_string(RT_TILE) + " pixels, the static scratch holds " + std::to_string(rt::kStaticTiles));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    rt::StaticScratch& sc = *ctx->staticScratch;
    // the context's own tables, then the array every context of this GPU shares: one caller at a time, and a caller on another
    // stream than the array's last user waits for the device first
    if (int rc2 = orderAfterOtherStreams(&ctx->lastStream, &ctx->lastStreamValid, stream)) return rc2;
    std::lock_guard<std::mutex> guard(sc.lock);
    if (int rc2 = orderAfterOtherStreams(&sc.lastStream, &sc.lastStreamValid, stream)) return rc2;
    if (ctx->sphere0Dirty || ctx->plane0Dirty) {
        RT_HIP(rt::launchPatchTables(ctx->sphere0Dirty ? const_cast<rt::F4*>(ctx->scene.sphGeom) : nullptr, ctx->sphere0Geom,
                                     ctx->plane0Dirty ? const_cast<rt::F4*>(ctx->scene.plnMat) : nullptr, ctx->plane0Mat0, stream));
        ctx->sphere0Dirty = ctx->plane0Dirty = false;
    }
    if (p.blockOrder != 2 || !ctx->costOrdering) {      // plain tile order: no scratch at all
        RT_HIP(rt::launchRender(ctx->scene, p, static_cast<uint8_t*>(d_rgb), nullptr, nullptr, nullptr, false, stream));
        return RT_OK;
    }
    // The array keeps what the last call on this GPU left in it: costs and order are only reused for the same frame geometry,
    // and both cost buffers start from zero otherwise.
    const long long key = planDispatch(ctx, p).key;
    const bool haveOrder = sc.orderKey == key;
    if (!haveOrder) RT_HIP(hipMemsetAsync(sc.words, 0, sizeof(uint32_t) * 2 * (size_t)rt::kStaticTiles, stream));
    uint32_t* cur = sc.words + (size_t)sc.costSlot * rt::kStaticTiles;
    uint32_t* nxt = sc.words + (size_t)(sc.costSlot ^ 1) * rt::kStaticTiles;
    uint32_t* order = sc.words + 2 * (size_t)rt::kStaticTiles;
    RT_HIP(rt::launchRender(ctx->scene, p, static_cast<uint8_t*>(d_rgb), nullptr, nullptr, nullptr, false, stream, nullptr,
                            haveOrder ? order : nullptr, cur));
    RT_HIP(rt::launchRankTiles(cur, order, nxt, p.ownedTiles, stream));
    sc.costSlot ^= 1;
    sc.orderKey = key;
    return RT_OK;
}

namespace {

// Camera-path launch: the cameras go to the device through a pinned staging buffer (stream-ordered copy), then one launch
// renders every frame; tile costs of all frames accumulate (maximum per tile) into the order used by the next launch.
int renderPath(rt_ctx* ctx, int width, int height, int rank, int world, bool compact, int nFrames, const rt_camera* cams, uint8_t* out,
               size_t frameBytes, hipStream_t stream) {
    if (nFrames <= 0 || nFrames > RT_MAX_PATH_FRAMES) return fail(RT_ERR_INVALID, "n_frames must be 1 .. " + std::to_string(RT_MAX_PATH_FRAMES));
    if (!cams) return fail(RT_ERR_INVALID, "null camera array");
    rt::FrameParams p = frameFor(ctx, width, height, rank, world, compact);
    if (p.ownedTiles <= 0) return RT_OK;
    if ((unsigned long long)rt::launchGridBlocks(p) * (unsigned long long)nFrames > 0x7fffffffull)
        return fail(RT_ERR_CAPACITY, "too many workgroups for one launch: render the path in shorter pieces");
    if (ctx->frame.meshMode == rt::kMeshNoBox && ctx->haveMeshBounds)
        for (int f = 0; f < nFrames; f++)
            if (!cameraWithinNoBoxRange(ctx, &cams[f].position.x))
                return fail(RT_ERR_INVALID, "boundingBox = false: camera " + std::to_string(f) + " of the path is more than 1500 units from the meshes");
    if (int rc = orderAfterOtherStreams(&ctx->lastStream, &ctx->lastStreamValid, stream)) return rc;
    const int slot = (int)(ctx->pathCalls++ % rt_ctx::kPathStaging);
    float* const staging = ctx->hPathCams + (size_t)slot * 16 * RT_MAX_PATH_FRAMES;
    RT_HIP(hipEventSynchronize(ctx->pathCamsFree[slot]));      // the copy of four path calls ago has left this staging slot
    for (int f = 0; f < nFrames; f++) {
        float* d = staging + 16 * (size_t)f;
        const rt_camera& c = cams[f];
        const float v[13] = {c.position.x, c.position.y, c.position.z, c.up.x, c.up.y, c.up.z, c.right.x, c.right.y, c.right.z,
                             c.direction.x, c.direction.y, c.direction.z, c.fovFactor};
        std::memcpy(d, v, sizeof v);
        // the view vector of the specular term, as rt_update_frame_state computes it (kernels.cu:48,56)
        const float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        d[13] = v[0] / len; d[14] = v[1] / len; d[15] = v[2] / len;
    }
    RT_HIP(hipMemcpyAsync(ctx->dPathCams, staging, sizeof(float) * 16 * (size_t)nFrames, hipMemcpyHostToDevice, stream));
    RT_HIP(hipEventRecord(ctx->pathCamsFree[slot], stream));
    if (ctx->sphere0Dirty || ctx->plane0Dirty) {
        RT_HIP(rt::launchPatchTables(ctx->sphere0Dirty ? const_cast<rt::F4*>(ctx->scene.sphGeom) : nullptr, ctx->sphere0Geom,
                                     ctx->plane0Dirty ? const_cast<rt::F4*>(ctx->scene.plnMat) : nullptr, ctx->plane0Mat0, stream));
        ctx->sphere0Dirty = ctx->plane0Dirty = false;
    }
    const DispatchPlan plan = planDispatch(ctx, p);
    if (!plan.ordered) {
        RT_HIP(rt::launchRenderPath(ctx->scene, p, out, frameBytes, nFrames, ctx->dPathCams, stream, nullptr, nullptr));
        return RT_OK;
    }
    // same two-buffer scheme as launchOrdered.  Nothing is split here, but a change of frame geometry must also forget the
    // blocks chosen for the old one: a later single-frame launch of this geometry would otherwise find an order in place and
    // trust a split list that indexes another frame's blocks.
    if (!plan.haveOrder) {
        RT_HIP(hipMemsetAsync(ctx->dTileCost, 0, sizeof(uint32_t) * kMaxOrderedTiles * 2, stream));
        const int blocks = p.ownedTiles * (RT_TILE / 8) * (RT_TILE / 8);
        if (ctx->split > 0 && blocks <= rt::kMaxSplitBlocks) {
            RT_HIP(hipMemsetAsync(ctx->dBlockCost, 0, sizeof(uint32_t) * (size_t)blocks, stream));
            RT_HIP(hipMemsetAsync(ctx->dSplitState, 0, sizeof(rt::SplitState), stream));
        }
        ctx->framesSinceSelection = 0;
    }
    uint32_t* cur = ctx->dTileCost + (size_t)ctx->costSlot * kMaxOrderedTiles;
    uint32_t* nxt = ctx->dTileCost + (size_t)(ctx->costSlot ^ 1) * kMaxOrderedTiles;
    RT_HIP(rt::launchRenderPath(ctx->scene, p, out, frameBytes, nFrames, ctx->dPathCams, stream, plan.haveOrder ? ctx->dTileOrder : nullptr, cur));
    RT_HIP(rt::launchRankTiles(cur, ctx->dTileOrder, nxt, p.ownedTiles, stream));
    ctx->costSlot ^= 1;
    ctx->orderKey = plan.key;
    return RT_OK;
}

}  // namespace

int rt_render_path_device(rt_ctx* ctx, int width, int height, int n_frames, const rt_camera* cams, void* d_rgb_frames, void* hip_stream) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (!d_rgb_frames) return fail(RT_ERR_INVALID, "null device buffer");
    return renderPath(ctx, width, height, 0, 1, false, n_frames, cams, static_cast<uint8_t*>(d_rgb_frames), (size_t)width * (size_t)height * 3,
                      static_cast<hipStream_t>(hip_stream));
}

int rt_render_path_tiles_device(rt_ctx* ctx, int width, int height, int rank, int world, int n_frames, const rt_camera* cams, void* d_tiles,
                                size_t frame_stride_bytes, void* hip_stream) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "rank/world out of range");
    if (!d_tiles) return fail(RT_ERR_INVALID, "null device buffer");
    if (frame_stride_bytes < tileBufferBytes(ctx, width, height, world) || frame_stride_bytes % 16 != 0)
        return fail(RT_ERR_INVALID, "frame stride must be a multiple of 16 and hold a tile buffer");
    return renderPath(ctx, width, height, rank, world, true, n_frames, cams, static_cast<uint8_t*>(d_tiles), frame_stride_bytes,
                      static_cast<hipStream_t>(hip_stream));
}

int rt_render_device(rt_ctx* ctx, int width, int height, void* d_rgb, void* hip_stream) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (!d_rgb) return fail(RT_ERR_INVALID, "null device buffer");
    rt::FrameParams p = frameFor(ctx, width, height, 0, 1, false);
    return launchOrdered(ctx, p, static_cast<uint8_t*>(d_rgb), nullptr, nullptr, nullptr, false, static_cast<hipStream_t>(hip_stream));
}

size_t rt_tile_buffer_bytes_dealt(int width, int height, int world, int rank0_slots, int other_slots) {
    if (width <= 0 || height <= 0 || world <= 0 || !validDeal(rank0_slots, other_slots)) return 0;
    const rt::TileDeal deal{world, rank0_slots, other_slots};
    // every rank's buffer has the size of the largest share
    return (size_t)rt::dealMostOwned(deal, tilesAlong(width) * tilesAlong(height)) * RT_TILE * RT_TILE * 3;
}

size_t rt_tile_buffer_bytes(int width, int height, int world) { return rt_tile_buffer_bytes_dealt(width, height, world, 1, 1); }

int rt_set_tile_deal(rt_ctx* ctx, int rank0_slots, int other_slots) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (!validDeal(rank0_slots, other_slots)) return fail(RT_ERR_INVALID, "slots per period must be 1 .. 64 for rank 0 and for the other ranks");
    ctx->dealRank0 = rank0_slots; ctx->dealOther = other_slots;
    return RT_OK;
}

int rt_get_tile_deal(rt_ctx* ctx, int* rank0_slots, int* other_slots) {
    if (!ctx || !rank0_slots || !other_slots) return fail(RT_ERR_INVALID, "null argument");
    *rank0_slots = ctx->dealRank0; *other_slots = ctx->dealOther;
    return RT_OK;
}

int rt_render_tiles_device(rt_ctx* ctx, int width, int height, int rank, int world, void* d_tiles, size_t capacity_bytes, void* hip_stream) {
    int rc = checkFrameArgs(ctx, width, height);
    if (rc) return rc;
    if (world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "rank/world out of range");
    if (!d_tiles) return fail(RT_ERR_INVALID, "null device buffer");
    if (capacity_bytes < tileBufferBytes(ctx, width, height, world)) return fail(RT_ERR_CAPACITY, "tile buffer too small");
    rt::FrameParams p = frameFor(ctx, width, height, rank, world, true);
    return launchOrdered(ctx, p, static_cast<uint8_t*>(d_tiles), nullptr, nullptr, nullptr, false, static_cast<hipStream_t>(hip_stream));
}

int rt_untile_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered, void* d_rgb, void* hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (width <= 0 || height <= 0 || world <= 0) return fail(RT_ERR_INVALID, "bad frame or world size");
    if (!d_gathered || !d_rgb) return fail(RT_ERR_INVALID, "null device buffer");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(rt::launchUntile(static_cast<const uint8_t*>(d_gathered), static_cast<uint8_t*>(d_rgb), width, height, dealOf(ctx, world), tilesAlong(width),
                            tileBufferBytes(ctx, width, height, world), static_cast<hipStream_t>(hip_stream)));
    return RT_OK;
}

int rt_untile_strided_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered, size_t rank_stride_bytes, void* d_rgb,
                             void* hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (width <= 0 || height <= 0 || world <= 0) return fail(RT_ERR_INVALID, "bad frame or world size");
    if (!d_gathered || !d_rgb) return fail(RT_ERR_INVALID, "null device buffer");
    if (rank_stride_bytes < tileBufferBytes(ctx, width, height, world) || rank_stride_bytes % 16 != 0)
        return fail(RT_ERR_INVALID, "rank stride must be a multiple of 16 and hold a tile buffer");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(rt::launchUntile(static_cast<const uint8_t*>(d_gathered), static_cast<uint8_t*>(d_rgb), width, height, dealOf(ctx, world), tilesAlong(width),
                            rank_stride_bytes, static_cast<hipStream_t>(hip_stream)));
    return RT_OK;
}

int rt_untile_batch_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered, size_t rank_stride_bytes,
                           size_t frame_stride_bytes, int n_frames, void* d_rgb_frames, void* hip_stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "null context");
    if (width <= 0 || height <= 0 || world <= 0) return fail(RT_ERR_INVALID, "bad frame or world size");
    if (n_frames <= 0 || n_frames > 65535) return fail(RT_ERR_INVALID, "n_frames must be 1 .. 65535");
    if (!d_gathered || !d_rgb_frames) return fail(RT_ERR_INVALID, "null device buffer");
    const size_t tileBytes = tileBufferBytes(ctx, width, height, world);
    if (frame_stride_bytes < tileBytes || frame_stride_bytes % 16 != 0) return fail(RT_ERR_INVALID, "frame stride must be a multiple of 16 and hold a tile buffer");
    if (rank_stride_bytes % 16 != 0 || rank_stride_bytes < frame_stride_bytes * (size_t)(n_frames - 1) + tileBytes)
        return fail(RT_ERR_INVALID, "rank stride must be a multiple of 16 and hold the tile buffers of all frames");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(rt::launchUntile(static_cast<const uint8_t*>(d_gathered), static_cast<uint8_t*>(d_rgb_frames), width, height, dealOf(ctx, world), tilesAlong(width),
                            rank_stride_bytes, static_cast<hipStream_t>(hip_stream), n_frames, frame_stride_bytes));
    return RT_OK;
}

int rt_get_scene_info(rt_ctx* ctx, rt_scene_info* info) {
    if (!ctx || !info) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->haveScene) return fail(RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    *info = ctx->info;
    return RT_OK;
}

int rt_build_octree(const rt_model* model, int max_depth, int leaf_triangles, rt_flat_octree* out) {
    if (!model || !out) return fail(RT_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    out->maxDepth = max_depth; out->leafTriangles = leaf_triangles;
    // a scene of this one model: its root is node 0, the other nodes follow breadth-first, triangle records start at 0
    rt_model alone = *model;
    alone.octree = nullptr;
    const rt_light light = {{0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}};
    rt_scene_desc scene;
    std::memset(&scene, 0, sizeof scene);
    scene.models = &alone; scene.modelCount = 1;
    scene.lights = &light; scene.lightCount = 1;
    scene.octreeMaxDepth = max_depth; scene.octreeLeafTriangles = leaf_triangles;
    rt::FlatScene flat;
    std::string err;
    if (!rt::flattenScene(scene, flat, err)) return fail(RT_ERR_INVALID, err);
    if (flat.forestRoot == rt::kNoRoot) return RT_OK;
    const size_t nodeBytes = flat.nodes.size() * sizeof(rt::F4);
    const size_t triBytes = (size_t)flat.triangleCount * 3 * sizeof(rt::F4);     // without the spare records at the end
    void* n = std::malloc(nodeBytes);
    void* t = std::malloc(triBytes ? triBytes : 1);
    if (!n || !t) { std::free(n); std::free(t); return fail(RT_ERR_INVALID, "out of host memory"); }
    std::memcpy(n, flat.nodes.data(), nodeBytes);
    std::memcpy(t, flat.tris.data(), triBytes);
    out->nodes = n; out->nodeCount = flat.nodeCount;
    out->triangles = t; out->triangleRecords = flat.triangleCount;
    out->leafCount = (int32_t)flat.leafCount;
    out->depth = flat.maxDepth;
    return RT_OK;
}

void rt_free_octree(rt_flat_octree* octree) {
    if (!octree) return;
    std::free(const_cast<void*>(octree->nodes));
    std::free(const_cast<void*>(octree->triangles));
    std::memset(octree, 0, sizeof *octree);
}

int rt_validate_scene_host(const rt_scene_desc* scene, rt_scene_info* info) {
    if (!scene) return fail(RT_ERR_INVALID, "null argument");
    rt::FlatScene flat;
    std::string err;
    if (!rt::flattenScene(*scene, flat, err)) return fail(RT_ERR_INVALID, err);
    if (!rt::validateFlatScene(*scene, flat, err)) return fail(RT_ERR_INVALID, err);
    if (info) {
        info->sphereCount = scene->sphereCount; info->planeCount = scene->planeCount; info->AABBCount = scene->AABBCount;
        info->modelCount = scene->modelCount; info->triangleCount = flat.triangleCount; info->nodeCount = flat.nodeCount;
        info->leafCount = flat.leafCount; info->maxDepth = flat.maxDepth;
        info->deviceBytes = (int64_t)((flat.sphGeom.size() + flat.sphMat.size() + flat.plnP.size() + flat.plnN.size() + flat.plnMat.size() + flat.plnNormals.size() +
                                       flat.boxMin.size() + flat.boxMax.size() + flat.boxMat.size() + flat.mdlGateMin.size() +
                                       flat.mdlGateMax.size() + flat.mdlTightMin.size() + flat.mdlTightMax.size() + flat.mdlMat.size() +
                                       flat.nodes.size() + flat.tris.size() + flat.attrs.size()) * sizeof(rt::F4) +
                                      flat.mdlInfo.size() * sizeof(uint32_t));
    }
    return RT_OK;
}

}  // extern "C"
