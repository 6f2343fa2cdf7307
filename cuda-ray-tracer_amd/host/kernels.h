// Host mirror of the reference's kernel entry point (kernels.h:11).  There the symbol is a __global__ function
// launched with <<<grid, block>>>; here it is a host function that forwards to the C ABI, which launches the gfx950
// kernels.  setupCurand (kernels.h:12) has no counterpart: the reference's random states are never consumed.
#pragma once
#include "RayTracer.h"

// Renders `scene` under `config` into framebuffer (width * height * 3 bytes).  sceneChanged: the scene arrays were
// replaced or resized since the previous call (a scene switch or an add*); the per-frame mutable members (light,
// spheres[0], planes[0] ambient colour, camera, config) are always re-read.  Returns an rt_status.
int rayTrace(rt_ctx* ctx, int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config,
             bool sceneChanged, rt_frame_stats* stats = nullptr);
