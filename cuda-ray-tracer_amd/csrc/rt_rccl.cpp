// Tile exchange over RCCL for frames sharded across the GPUs of a node (see include/rt_rccl.h).
#include "rt_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "identifier sizes must agree");

struct rt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    uint8_t* dTiles = nullptr; size_t tilesCap = 0;
    uint8_t* dGathered = nullptr; size_t gatheredCap = 0;     // rank 0 only
    uint8_t* dFrame = nullptr; size_t frameCap = 0;           // rank 0 only
    int* dStatus = nullptr;                                   // two ints: this rank's status of the frame, the maximum over ranks
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define RT_TRY_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define RT_TRY_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(RT_ERR_DEVICE, std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)

int ensure(uint8_t** buf, size_t* cap, size_t need) {
    if (*cap >= need) return RT_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    void* p = nullptr;
    RT_TRY_HIP(hipMalloc(&p, need));
    *buf = static_cast<uint8_t*>(p); *cap = need;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_comm_last_error(void) { return g_error.c_str(); }

int rt_comm_make_id(unsigned char id[RT_COMM_ID_BYTES]) {
    if (!id) return fail(RT_ERR_INVALID, "null identifier");
    ncclUniqueId u;
    RT_TRY_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_create(const unsigned char id[RT_COMM_ID_BYTES], int rank, int world, int device, rt_comm** out) {
    if (!id || !out) return fail(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "rank/world out of range");
    RT_TRY_HIP(hipSetDevice(device));
    rt_comm* c = new (std::nothrow) rt_comm();
    if (!c) return fail(RT_ERR_INVALID, "out of host memory");
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    std::memcpy(u.internal, id, RT_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return fail(RT_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { rt_comm_destroy(c); return fail(RT_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    void* st = nullptr;
    e = hipMalloc(&st, 2 * sizeof(int));
    if (e != hipSuccess) { rt_comm_destroy(c); return fail(RT_ERR_DEVICE, std::string("hipMalloc(status): ") + hipGetErrorString(e)); }
    c->dStatus = static_cast<int*>(st);
    *out = c;
    return RT_OK;
}

void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->dTiles) (void)hipFree(c->dTiles);
    if (c->dGathered) (void)hipFree(c->dGathered);
    if (c->dFrame) (void)hipFree(c->dFrame);
    if (c->dStatus) (void)hipFree(c->dStatus);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

int rt_gather_tiles(rt_comm* c, const void* d_tiles, size_t bytes_per_rank, void* d_gathered, void* hip_stream) {
    if (!c || !d_tiles) return fail(RT_ERR_INVALID, "null argument");
    if (c->rank == 0 && !d_gathered) return fail(RT_ERR_INVALID, "rank 0 needs a buffer for the gathered tiles");
    RT_TRY_HIP(hipSetDevice(c->device));
    RT_TRY_NCCL(ncclGather(d_tiles, d_gathered, bytes_per_rank, ncclChar, 0, c->comm, static_cast<hipStream_t>(hip_stream)));
    return RT_OK;
}

int rt_render_frame_sharded(rt_ctx* ctx, rt_comm* c, int width, int height, uint8_t* rgb_out) {
    if (!ctx || !c) return fail(RT_ERR_INVALID, "null argument");
    if (c->rank == 0 && !rgb_out) return fail(RT_ERR_INVALID, "rank 0 needs an output buffer");
    int rank0Slots = 1, otherSlots = 1;      // the context's deal of the tiles (rt_set_tile_deal), the same on every rank
    if (rt_get_tile_deal(ctx, &rank0Slots, &otherSlots) != RT_OK) return fail(RT_ERR_INVALID, rt_last_error());
    const size_t per = rt_tile_buffer_bytes_dealt(width, height, c->world, rank0Slots, otherSlots);
    if (per == 0) return fail(RT_ERR_INVALID, "bad frame size");
    RT_TRY_HIP(hipSetDevice(c->device));
    // What can fail on this rank alone (an allocation, the render call's own checks) must not keep it out of the collectives:
    // a rank that returned here would leave the others blocked in the gather for ever.  So every rank first takes part in a
    // one-word all-reduce of its status, and the frame's gather only happens if every rank is ready for it.
    int local = RT_OK;
    std::string localError;
    auto attempt = [&](int rc, const char* message) { if (local == RT_OK && rc != RT_OK) { local = rc; localError = message; } };
    attempt(ensure(&c->dTiles, &c->tilesCap, per), g_error.c_str());
    if (c->rank == 0) {
        attempt(ensure(&c->dGathered, &c->gatheredCap, per * (size_t)c->world), g_error.c_str());
        attempt(ensure(&c->dFrame, &c->frameCap, (size_t)width * (size_t)height * 3), g_error.c_str());
    }
    if (local == RT_OK && rt_render_tiles_device(ctx, width, height, c->rank, c->world, c->dTiles, per, c->stream) != RT_OK) attempt(RT_ERR_DEVICE, rt_last_error());
    int anyFailed = 0;
    RT_TRY_HIP(hipMemsetAsync(c->dStatus, local == RT_OK ? 0 : 1, sizeof(int), c->stream));
    RT_TRY_NCCL(ncclAllReduce(c->dStatus, c->dStatus + 1, 1, ncclInt, ncclMax, c->comm, c->stream));
    RT_TRY_HIP(hipMemcpyAsync(&anyFailed, c->dStatus + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RT_TRY_HIP(hipStreamSynchronize(c->stream));
    if (local != RT_OK) return fail(local, localError);
    if (anyFailed) return fail(RT_ERR_DEVICE, "another rank could not render its share of the frame");
    int rc = RT_OK;
    if ((rc = rt_gather_tiles(c, c->dTiles, per, c->dGathered, c->stream))) return rc;
    if (c->rank == 0) {
        if (rt_untile_device(ctx, width, height, c->world, c->dGathered, c->dFrame, c->stream) != RT_OK) return fail(RT_ERR_DEVICE, rt_last_error());
        RT_TRY_HIP(hipMemcpyAsync(rgb_out, c->dFrame, (size_t)width * (size_t)height * 3, hipMemcpyDeviceToHost, c->stream));
    }
    RT_TRY_HIP(hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // extern "C"
