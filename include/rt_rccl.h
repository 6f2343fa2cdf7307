/*
 * rt_rccl.h - exchange of image tiles between the GPUs of one node, for frames sharded with rt_render_tiles_device.
 *
 * One process per GPU; every process holds a context with the same scene; rank r renders tiles r, r + world, ... of a frame
 * and one gather over RCCL (xGMI: every peer has a link of its own to rank 0) brings the compact tile buffers to rank 0,
 * which reassembles the frame.  The reference is a single-GPU program (RayTracer.cpp:405-417 launches on the current
 * device only), so nothing here has a counterpart in it.
 *
 * Lives in a library of its own (librt_amd_rccl.so, which links librccl) so that librt_amd.so itself depends on nothing but
 * the HIP runtime.  Errors: non-zero rt_status and rt_comm_last_error().
 */
#ifndef RT_RCCL_H
#define RT_RCCL_H

#include <stddef.h>
#include <stdint.h>
#include "rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_comm rt_comm;

#define RT_COMM_ID_BYTES 128

/* Rank 0 makes the identifier of a new communicator (ncclGetUniqueId) and hands it to every rank (pipe, file, launcher). */
int rt_comm_make_id(unsigned char id[RT_COMM_ID_BYTES]);

/* Joins the communicator: collective over all `world` ranks (ncclCommInitRank).  `device` is the rank's HIP device. */
int rt_comm_create(const unsigned char id[RT_COMM_ID_BYTES], int rank, int world, int device, rt_comm** out);
void rt_comm_destroy(rt_comm* comm);

/*
 * Gathers bytes_per_rank bytes from every rank's d_tiles into d_gathered on rank 0 (rank order, bytes_per_rank apart;
 * ignored on the other ranks), enqueued on hip_stream: ncclGather.  Collective.
 */
int rt_gather_tiles(rt_comm* comm, const void* d_tiles, size_t bytes_per_rank, void* d_gathered, void* hip_stream);

/*
 * One sharded frame, blocking: this rank's tiles (rt_render_tiles_device), the gather, and on rank 0 the reassembly
 * (rt_untile_device) and the copy to rgb_out (host, 3 * width * height bytes; ignored on the other ranks).  Collective.
 * The multi-GPU form of rt_render_frame; device buffers are kept in the communicator and grown when a larger frame comes.
 * A failure on one rank (an allocation, the render call's own checks) is agreed on with a one-word all-reduce before the
 * gather, so every rank returns - the failing one with its own error, the others with RT_ERR_DEVICE - instead of waiting in
 * the collective for a rank that has left.
 */
int rt_render_frame_sharded(rt_ctx* ctx, rt_comm* comm, int width, int height, uint8_t* rgb_out);

const char* rt_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_RCCL_H */
