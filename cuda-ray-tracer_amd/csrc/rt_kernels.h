// Host-callable launchers of the gfx950 kernels (defined in rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "rt_device_types.h"

#define RT_TILE_PX 32   // must equal RT_TILE of the public header

namespace rt {

// Renders the tiles FrameParams says this launch owns.  `out` is either the full row-major frame or the rank's
// compact tile buffer (P.compactOutput).  ids / tbuf / counters may be null.  stats selects the instrumented kernel.
hipError_t launchRender(const DevScene& S, const FrameParams& P, uint8_t* out, int32_t* ids, float* tbuf,
                        unsigned long long* counters, bool stats, hipStream_t stream, unsigned long long* waveClocks = nullptr,
                        const uint32_t* tileOrder = nullptr, uint32_t* tileCost = nullptr, BlockFeedback fb = BlockFeedback{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});

// Renders `frames` frames of one geometry in one launch: frame f uses the camera dPathCams[16 f .. 16 f + 16) (device memory:
// position, up, right, direction, fovFactor, view vector) and writes to out + f * frameBytes; everything else comes from P.
hipError_t launchRenderPath(const DevScene& S, const FrameParams& P, uint8_t* out, size_t frameBytes, int frames, const float* dPathCams,
                            hipStream_t stream, const uint32_t* tileOrder, uint32_t* tileCost);

// Block feedback between frames (see BlockFeedback): gathers the statistics of the costs in blockCost into slot `slot` of
// `state`, chooses the blocks the next frame renders with split waves (extraMap, heavyBits, state->used[slot]) and clears
// blockCost.  slots = waves resident on the device at once (residentRenderWaves()).  heavyBits needs blocks / 32 + 8 words.
hipError_t launchSelectHeavy(uint32_t* blockCost, uint32_t* extraMap, uint32_t* heavyBits, int blocks, SplitState* state, int slot, int slots,
                             int ratio256, uint32_t floorCost, int startTenths, hipStream_t stream);
int residentRenderWaves();

// order = this rank's tiles sorted by descending cost (ties by index); clearNext[0..n) is zeroed for a later frame.
hipError_t launchRankTiles(const uint32_t* cost, uint32_t* order, uint32_t* clearNext, int n, hipStream_t stream);

// Workgroups (= waves) launchRender starts for these parameters.
unsigned launchGridBlocks(const FrameParams& P);

// Enqueues the rewrite of sphere 0's geometry record and / or plane 0's ambient colour (null pointer = leave alone).
hipError_t launchPatchTables(F4* sphGeom0, F4 sphere0, F4* plnMat0, F4 plane0Mat0, hipStream_t stream);

// Dispatch-feedback scratch that lives in a file-scope device array instead of an allocation (rt_render_device_static):
// words = this device's copy of the array: tile costs [0, kStaticTiles) and [kStaticTiles, 2 kStaticTiles), tile order after
// them.  costSlot / orderKey say which cost buffer the next frame accumulates into and which frame geometry the order is for.
constexpr int kStaticTiles = 32768;
// lock: held for the whole of a call that uses the array (host bookkeeping and the launches' enqueueing); lastStream: the stream
// those launches went to.
struct StaticScratch { uint32_t* words; int costSlot; long long orderKey; std::mutex lock; hipStream_t lastStream; bool lastStreamValid; };
// Looks the array's address up on first use per device (which must be the current device) and returns that device's record.
hipError_t staticScratch(int device, StaticScratch** out);

hipError_t launchUntile(const uint8_t* gathered, uint8_t* out, int width, int height, TileDeal deal, int tilesX,
                        size_t perRankBytes, hipStream_t stream, int frames = 1, size_t frameStrideIn = 0);

}  // namespace rt
