/*
 * rt_api.h — C ABI of the MI355X (gfx950) renderer.
 *
 * This is the drop-in boundary for the reference's render path: the sole
 * caller of the reference kernel is RayTracer::launchKernel()
 * (RayTracer.cpp:405-417), which launches kernels.h:11 `rayTrace` and then
 * synchronises.  A host application keeps its Scene / SceneConfig / Camera
 * objects, mirrors them into the plain structs of rt_types.h, and calls the
 * functions below instead of launching the kernel itself.
 *
 * Conventions
 *  - every function returns 0 on success and a non-zero rt_status otherwise;
 *    rt_last_error() then returns a message for the calling thread.  (The
 *    reference only prints cudaGetErrorString to stderr and carries on,
 *    RayTracer.cpp:411-414.)
 *  - inputs are caller-owned and only read during the call; device memory is
 *    owned by the context.
 *  - a context is not re-entrant: one thread at a time per rt_ctx, exactly as
 *    the reference's single GLFW thread drives one RayTracer.  A context keeps
 *    dispatch feedback between frames (tile costs and the order ranked from
 *    them), so its frames belong on ONE stream: a render call on another stream
 *    than the context's previous one (the blocking entry points use a stream of
 *    the context's own) first waits for the device to go idle.  For frames in
 *    flight on several streams give each stream a context of its own.
 *  - host-pointer entry points block until the frame is complete (the
 *    reference's cudaDeviceSynchronize, RayTracer.cpp:416); *_device entry
 *    points enqueue on the given HIP stream and return.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = 1,     /* bad argument / scene fails validation */
    RT_ERR_DEVICE = 2,      /* HIP runtime error */
    RT_ERR_NO_SCENE = 3,    /* render requested before rt_upload_scene */
    RT_ERR_NO_STATE = 4,    /* render requested before rt_update_frame_state */
    RT_ERR_CAPACITY = 5     /* caller buffer too small */
};

/* Image tiles used to shard a frame over GPUs: RT_TILE x RT_TILE pixels, tile i belongs to rank i % world. */
#define RT_TILE 32

/*
 * Creates a context on HIP device devices[0] (device_count must be 1: one process drives one GPU; a frame
 * is sharded over several GPUs by giving each process its own context and rank, see rt_render_tiles_device).
 * Replaces the device-side set-up in RayTracer::init / initialiseScenes (RayTracer.cpp:58-69, 342-351);
 * the curand state buffer of the reference (RayTracer.cpp:344-345, kernels.cu:366-369) is never consumed
 * by its kernel and has no counterpart here.
 */
int rt_create(int device_count, const int* devices, rt_ctx** out);

/* Frees every device allocation of the context.  Replaces RayTracer::~RayTracer (RayTracer.cpp:18-56). */
void rt_destroy(rt_ctx* ctx);

/*
 * Flattens a host scene into device-resident SoA buffers and builds the per-model octrees.
 * Replaces the managed-memory scene arrays filled by RayTracer::initialiseScenes (RayTracer.cpp:71-337),
 * RayTracer::switchScene (RayTracer.cpp:356-358), addSphere/addPlane/addAABB (RayTracer.cpp:360-403) and the
 * upload half of Model::Model (Model.cpp:55-56, 70-117).  Call again whenever objects are added or replaced.
 */
int rt_upload_scene(rt_ctx* ctx, const rt_scene_desc* scene);

/*
 * Replaces the spheres, planes, boxes and the light of the uploaded scene by those of `scene` and leaves the uploaded models -
 * their triangles, attributes and octrees - where they are (scene->models is not read).  This is what addSphere / addPlane /
 * addAABB (RayTracer.cpp:360-403: reallocate the one table, copy, append) cost in the reference; a full rt_upload_scene after
 * such a call would flatten every mesh again.  Blocks until launches that read the old tables have finished.  On failure the
 * context keeps the scene it had.
 */
int rt_update_objects(rt_ctx* ctx, const rt_scene_desc* scene);

/*
 * Sets what the reference passes by value on every launch (Scene::cam and SceneConfig, kernels.h:11) plus the
 * scene members its frame loop rewrites between launches (Window.cpp:357-384).  dyn may be NULL.
 */
int rt_update_frame_state(rt_ctx* ctx, const rt_camera* cam, const rt_config* config, const rt_dynamic* dyn);

/*
 * What config.octree = true means.  In the reference that branch of Model::hit is an unfinished stub that returns false
 * (Model.cpp:163-219), so the flag hides every mesh and its shadows.  By default the library deviates: the flag selects the
 * whole-model box gate with the octree as accelerator, i.e. meshes render as with boundingBox = true.  strict != 0 restores
 * the reference's behaviour for this context (meshes invisible while the flag is set).  Takes effect with the next
 * rt_update_frame_state.
 */
int rt_set_strict_octree_flag(rt_ctx* ctx, int strict);

/*
 * Renders one width x height frame and copies it to rgb_out (host, 3*width*height bytes, row-major, RGB8).
 * Replaces RayTracer::launchKernel (RayTracer.cpp:405-417) plus the host read of the managed framebuffer
 * (Window.cpp:414).  stats may be NULL.
 */
int rt_render_frame(rt_ctx* ctx, int width, int height, uint8_t* rgb_out, rt_frame_stats* stats);

/*
 * As rt_render_frame, and also returns per-pixel primary-hit identities: ids_out holds two int32 per pixel,
 * (objectName << 24 | object index, triangle index within the model) or (-1, -1) for a miss; t_out the
 * primary hit distance (999 = miss).  Either may be NULL.  The reference keeps no such buffers; they exist
 * so that hit parity can be checked.  The launch also fills the traversal counters of rt_frame_stats.
 */
int rt_render_frame_ids(rt_ctx* ctx, int width, int height, uint8_t* rgb_out, int32_t* ids_out, float* t_out,
                        rt_frame_stats* stats);

/*
 * Enqueues one frame on hip_stream (a hipStream_t, NULL = default stream) writing RGB8 to the device buffer
 * d_rgb (3*width*height bytes).  The kernels.h:11 launch without the synchronise.  Allocates nothing.
 */
int rt_render_device(rt_ctx* ctx, int width, int height, void* d_rgb, void* hip_stream);

/*
 * As rt_render_device, for callers that cannot allocate device memory and cannot rely on memory a context allocated:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * The dispatch feedback of the frame (tile costs of the previous frame and the tile order derived from them) lives in a
 * file-scope device array of the library instead of the context's buffers: 3 x 32768 32-bit words = 384 KiB per GPU, held
 * by every GPU the library runs on until the process ends.  Frames of more than 32768 tiles of RT_TILE x RT_TILE pixels
 * (7680 x 4320 has 32400) are rejected with RT_ERR_CAPACITY.  The call itself calls no hipMalloc / hipFree or allocating
 * routine; the scene must have been uploaded and the frame state set before.  Heavy blocks are not split in this variant.
 * Every call on a GPU shares that GPU's array, whichever context it comes through.  Calls are serialised on the host by a
 * lock, and a call on another stream than the array's previous user first waits for the device to go idle (the order list must
 * not be read or ranked while another stream's launches still use it: a torn list would leave tiles unrendered), so frames
 * issued through this entry point on several streams of one GPU do not overlap; use one stream per GPU for it.  The array
 * keeps the last call's contents; a call for another frame geometry clears the costs.
 */
int rt_render_device_static(rt_ctx* ctx, int width, int height, void* d_rgb, void* hip_stream);

/*
 * Renders n_frames frames of the uploaded scene along a camera path with ONE launch: frame f uses cams[f] (host array, read
 * during the call) and everything else from the current frame state (configuration, light, sphere 0, floor colour), and is
 * written to d_rgb_frames + f * 3 * width * height.  This is the reference's frame loop with a moving camera
 * (Window.cpp:215-326, 338-421: camera moves, then launchKernel, every frame) folded into one enqueue: consecutive frames
 * follow each other in dispatch order, so the first waves of a frame fill the GPU while the slow tail of the frame before it
 * drains, and the host issues one launch per path instead of one per frame.  Byte for byte the frames rt_render_device
 * produces one at a time.  At most RT_MAX_PATH_FRAMES frames per call.  Allocates nothing.
 */
#define RT_MAX_PATH_FRAMES 1024
int rt_render_path_device(rt_ctx* ctx, int width, int height, int n_frames, const rt_camera* cams, void* d_rgb_frames, void* hip_stream);

/*
 * The same for one rank of an image-tile sharded run: frame f's tiles (layout of rt_render_tiles_device) are written to
 * d_tiles + f * frame_stride_bytes; frame_stride_bytes >= rt_tile_buffer_bytes(width, height, world), a multiple of 16.
 */
int rt_render_path_tiles_device(rt_ctx* ctx, int width, int height, int rank, int world, int n_frames, const rt_camera* cams,
                                void* d_tiles, size_t frame_stride_bytes, void* hip_stream);

/* Bytes one rank must provide to rt_render_tiles_device for a width x height frame split over `world` ranks. */
size_t rt_tile_buffer_bytes(int width, int height, int world);

/*
 * How the tiles of a sharded frame are dealt to the ranks, a setting of the context that the tile entry points below read
 * (rt_render_tiles_device, rt_render_path_tiles_device, rt_untile_device, rt_untile_strided_device, rt_untile_batch_device):
 * tiles are dealt in periods of rank0_slots + (world - 1) * other_slots consecutive tiles, of which the first rank0_slots go
 * to rank 0, the next other_slots to rank 1, and so on; a rank's buffer holds its tiles in frame order.  The default, 1 / 1,
 * is tile i -> rank i % world.  Rank 0 reassembles every frame besides rendering its share: with 4 / 5 on eight GPUs it
 * renders 4/39 of the tiles instead of 1/8 and the others 5/39 each.  Every rank of a run, and the context that reassembles,
 * must use the same deal; buffers then need rt_tile_buffer_bytes_dealt bytes (the largest share).  Slots: 1 .. 64.
 * No reference counterpart (the reference is single-GPU).
 */
int rt_set_tile_deal(rt_ctx* ctx, int rank0_slots, int other_slots);
int rt_get_tile_deal(rt_ctx* ctx, int* rank0_slots, int* other_slots);
size_t rt_tile_buffer_bytes_dealt(int width, int height, int world, int rank0_slots, int other_slots);

/*
 * Renders only the RT_TILE x RT_TILE tiles owned by `rank` (tile i, row-major over the tile grid, belongs to
 * rank i % world, or to the rank that rt_set_tile_deal's periods give it) into d_tiles, compactly: the rank's j-th tile occupies bytes [j*3*RT_TILE^2, (j+1)*3*RT_TILE^2),
 * row-major inside the tile, edge tiles padded.  No reference counterpart (the reference is single-GPU).
 */
int rt_render_tiles_device(rt_ctx* ctx, int width, int height, int rank, int world, void* d_tiles,
                           size_t capacity_bytes, void* hip_stream);

/*
 * Reassembles a frame from `world` tile buffers laid end to end in d_gathered (each rt_tile_buffer_bytes long,
 * rank order) into the row-major RGB8 frame d_rgb.  Runs on the context's device.
 */
int rt_untile_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered, void* d_rgb,
                     void* hip_stream);

/*
 * The same with the tile buffers `rank_stride_bytes` apart (at least rt_tile_buffer_bytes, a multiple of 16): for
 * callers that gather the tiles of several frames with one collective, rank r's buffer of the frame then starts at
 * d_gathered + r * rank_stride_bytes.  No reference counterpart.
 */
int rt_untile_strided_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered,
                             size_t rank_stride_bytes, void* d_rgb, void* hip_stream);

/*
 * The frames of a whole gather batch with one launch: frame j of rank r starts at d_gathered + r * rank_stride_bytes +
 * j * frame_stride_bytes (both multiples of 16), and the reassembled frames are written one after the other into
 * d_rgb_frames (n_frames x width x height x 3 bytes).  One launch instead of n: beside render launches that occupy the
 * GPU a reassembly kernel mostly waits for registers to come free, once per launch.  No reference counterpart.
 */
int rt_untile_batch_device(rt_ctx* ctx, int width, int height, int world, const void* d_gathered,
                           size_t rank_stride_bytes, size_t frame_stride_bytes, int n_frames, void* d_rgb_frames,
                           void* hip_stream);

/* Ray count of the current scene/state at this resolution, from the device's own counters (blocking). */
int rt_count_rays(rt_ctx* ctx, int width, int height, rt_frame_stats* stats);

/*
 * Diagnostic: renders one frame with the instrumented kernel and returns RT_WAVE_RECORD_WORDS 64-bit words per wave of
 * the launch: [0],[1] the 100 MHz wall-clock ticks at which it started and finished (0,0 for waves of padding
 * workgroups); then (shader cycles, executions + lanes active at those executions << 32) pairs for eight sections: octree node visits, triangle tests, the
 * up-front model gates, whole mesh traces, linear-table closest traces, surface + lighting, linear-table shadow tests,
 * unused.  Wave i is workgroup i of the launch.  capacity_pairs counts waves (records).  *waves_out receives the wave
 * count.  No reference counterpart.
 */
#define RT_WAVE_RECORD_WORDS 18
int rt_profile_waves(rt_ctx* ctx, int width, int height, uint64_t* clocks_out, size_t capacity_pairs, size_t* waves_out);

/* Sizes of the flattened scene (blocking, host data only). */
int rt_get_scene_info(rt_ctx* ctx, rt_scene_info* info);

/*
 * Host only (no GPU): flattens `scene` exactly as rt_upload_scene would and checks the octree invariants the
 * traversal relies on (every reachable triangle in exactly one leaf and inside its padded box, nested child boxes,
 * contiguous children, depth within the traversal stack).  Fills `info` (may be NULL).  No reference counterpart.
 */
int rt_validate_scene_host(const rt_scene_desc* scene, rt_scene_info* info);

/*
 * Copies the node and triangle arrays of the uploaded scene back from the GPU and runs the same checks on them against
 * `scene`, which must be the description the context was uploaded from.  This is how a GPU-built octree
 * (rt_scene_desc.buildOnDevice) is verified.  Blocking.  No reference counterpart.
 */
int rt_validate_scene_device(rt_ctx* ctx, const rt_scene_desc* scene);

/* Message describing the last failure on this thread ("" if none). */
const char* rt_last_error(void);

/* ---- scene-ingest helpers (host only; no GPU needed) ---- */

/*
 * Builds the octree of one model on the host, as the flat arrays the kernels read: Octree::init + Octree::build
 * (Octree.cpp:7-43), redefined as a spatial build (the reference's own build deals the vertex list out sequentially and
 * is never called).  max_depth / leaf_triangles: 0 = library default.  The arrays belong to `out` until rt_free_octree.
 * A model without reachable triangles yields nodeCount = 0.  The result may be passed back in rt_model.octree of the
 * same model (same vertices, indices and chunk boxes).
 */
int rt_build_octree(const rt_model* model, int max_depth, int leaf_triangles, rt_flat_octree* out);
void rt_free_octree(rt_flat_octree* octree);

/*
 * Camera basis from position, fov, yaw and pitch in degrees: Camera::Camera + Camera::updateDirection
 * (Camera.cpp:5-20, 54-63).
 */
int rt_camera_look(const float position[3], float fov, float yaw, float pitch, rt_camera* out);

/*
 * Bakes loader output into a world-space model: the body of Model::Model (Model.cpp:62-126).  Inputs are per
 * loaded vertex (LoadedVertices order): pos/nrm/ka/kd/ks 3 floats each, tex 2 floats; loaded_indices as
 * produced by the OBJ loader.  Outputs: vertices[vert_count], indices[indices_count], bb[6] (whole-model box
 * min then max), chunk_boxes[48] (eight boxes, min then max), chunk_indices[8 * *indices_per_box] where
 * the caller provides room for 8 * 3 * ceil((indices_count / 3) / 8.0) ints.
 */
int rt_bake_model(const float* pos, const float* nrm, const float* tex, const float* ka, const float* kd,
                  const float* ks, int vert_count, const uint32_t* loaded_indices, int indices_count,
                  const float position[3], int size, rt_vertex* vertices, int32_t* indices, float* bb,
                  float* chunk_boxes, int32_t* chunk_indices, int32_t* indices_per_box);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
