// Host mirror of the reference's kernel entry point (kernels.h:11).  There the symbol is a __global__ function
// launched with <<<grid, block>>>; here it is a host function that forwards to the C ABI, which launches the gfx950
// kernels.  setupCurand (kernels.h:12) has no counterpart: the reference's random states are never consumed.
#pragma once
#include "RayTracer.h"

// The reference's per-thread random state (curand_kernel.h).  Its kernel loads and stores one per pixel but never draws a
// number from it (kernels.cu:326-327, 359; every curand call is commented out, :289-292), so the type stays opaque here.
struct curandState;

// kernels.h:11, argument for argument: width, height, framebuffer (width * height * 3 bytes, host memory), the scene and the
// configuration, and the random states, which are ignored.  (The reference takes Scene and SceneConfig by value because they
// are kernel arguments; a host function binds them by reference - call sites read the same.)  Renders on a process-wide
// default context (GPU RT_DEVICE, default 0), created at the first call; the scene arrays are uploaded at the first call and
// again whenever the tables or a model's arrays differ from what was uploaded.  Blocking, like the launch followed by
// cudaDeviceSynchronize in RayTracer::launchKernel (RayTracer.cpp:405-417).  Errors are printed to std::cerr and the call
// returns, which is the reference's policy (RayTracer.cpp:411-414); rayTraceStatus() returns the rt_status of the last call.
void rayTrace(int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config, curandState* randStates);
int rayTraceStatus();
// kernels.h:12.  Nothing to set up: no random state is ever consumed.
void setupCurand(curandState* randStates, unsigned long seed);
// Destroys the default context (optional; for orderly shutdown before the HIP runtime unloads).
void rayTraceShutdown();

// What rayTrace does before it renders: uploads the scene arrays if they changed and hands over everything that travels with
// a frame (camera, configuration, light, spheres[0], planes[0] ambient colour).  Returns an rt_status.
int prepareFrame(rt_ctx* ctx, const Scene& scene, const SceneConfig& config, SceneChange sceneChanged);

// Renders `scene` under `config` into framebuffer (width * height * 3 bytes).  sceneChanged: how the scene arrays were
// replaced or resized since the previous call (a scene switch or an add*); the per-frame mutable members (light,
// spheres[0], planes[0] ambient colour, camera, config) are always re-read.  Returns an rt_status.
int rayTrace(rt_ctx* ctx, int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config,
             SceneChange sceneChanged, rt_frame_stats* stats = nullptr);
