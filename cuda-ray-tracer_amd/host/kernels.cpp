#include "kernels.h"

int rayTrace(rt_ctx* ctx, int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config,
             bool sceneChanged, rt_frame_stats* stats) {
    // the scene arrays live in HBM; they are re-flattened only when the host replaced them
    if (sceneChanged) {
        SceneDescStorage storage;
        rt_scene_desc desc;
        describeScene(scene, desc, storage);
        int rc = rt_upload_scene(ctx, &desc);
        if (rc != RT_OK) return rc;
    }
    rt_camera cam = scene.cam.toRt();
    rt_config cfg = toRtConfig(config);
    rt_dynamic dyn{};
    auto rv = [](const vec3& v) { return rt_vec3{v[0], v[1], v[2]}; };
    if (!scene.lights.empty()) { dyn.setLight = 1; dyn.lightPosition = rv(scene.lights[0].position); }
    if (!scene.spheres.empty()) { dyn.setSphere0 = 1; dyn.sphere0Position = rv(scene.spheres[0].position); dyn.sphere0Radius = scene.spheres[0].radius; }
    if (!scene.planes.empty()) { dyn.setPlane0 = 1; dyn.plane0AmbientColour = rv(scene.planes[0].mat.ambientColour); }
    int rc = rt_update_frame_state(ctx, &cam, &cfg, &dyn);
    if (rc != RT_OK) return rc;
    return rt_render_frame(ctx, width, height, framebuffer, stats);
}
