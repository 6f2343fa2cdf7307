#include "kernels.h"

#include <cstdlib>
#include <cstring>
#include <iostream>

namespace {

rt_ctx* g_defaultCtx = nullptr;
unsigned long long g_uploadedObjects = 0, g_uploadedMeshes = 0;     // fingerprints of the scene the default context holds
int g_lastStatus = RT_OK;

void mix(unsigned long long& h, const void* data, size_t bytes) {      // FNV-1a
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
}

// Everything rt_upload_scene reads, except what travels with every frame anyway (camera, light position, spheres[0] geometry,
// planes[0] ambient colour), as two fingerprints: the tables of spheres, planes and boxes with the light's colour (what
// rt_update_objects replaces), and the models with the octree build parameters (which only a full upload replaces).  Models are
// identified by their arrays (address and size), their placement and a sample of their vertices: the reference never edits a
// model's vertices after construction, but a model freed and rebuilt may land on the same addresses.
unsigned long long objectsFingerprint(rt_scene_desc desc, SceneDescStorage storage) {
    unsigned long long h = 1469598103934665603ull;
    if (!storage.spheres.empty()) { storage.spheres[0].position = rt_vec3{0, 0, 0}; storage.spheres[0].radius = 0; }
    if (!storage.planes.empty()) storage.planes[0].mat.ambientColour = rt_vec3{0, 0, 0};
    const int counts[3] = {desc.sphereCount, desc.planeCount, desc.AABBCount};
    mix(h, counts, sizeof counts);
    for (const rt_light& l : storage.lights) mix(h, &l.colour, 12);
    // field by field where a struct has padding
    for (const rt_sphere& s : storage.spheres) { mix(h, &s.position, 12); mix(h, &s.radius, 4); mix(h, &s.mat, sizeof s.mat); mix(h, &s.objectType, 8); }
    for (const rt_plane& p : storage.planes) { mix(h, &p.position, 24); mix(h, &p.mat, sizeof p.mat); mix(h, &p.objectType, 8); }
    for (const rt_aabb& b : storage.boxes) { mix(h, &b.min, 24); mix(h, &b.mat, sizeof b.mat); mix(h, &b.objectType, 8); }
    return h ? h : 1ull;
}

unsigned long long meshesFingerprint(const rt_scene_desc& desc, const SceneDescStorage& storage) {
    unsigned long long h = 1469598103934665603ull;
    const int counts[4] = {desc.modelCount, desc.octreeMaxDepth, desc.octreeLeafTriangles, desc.buildOnDevice};
    mix(h, counts, sizeof counts);
    for (const rt_model& m : storage.models) {
        mix(h, &m.position, 12); mix(h, &m.size, 4); mix(h, &m.mat, sizeof m.mat); mix(h, &m.objectType, 8);
        mix(h, &m.bbMin, 24); mix(h, &m.vertices, sizeof m.vertices); mix(h, &m.vertCount, 4);
        mix(h, &m.indices, sizeof m.indices); mix(h, &m.indicesCount, 4); mix(h, &m.octree, sizeof m.octree);
        if (m.vertices && m.vertCount > 0)      // 64 vertices spread over the array
            for (int k = 0; k < 64; k++) mix(h, &m.vertices[(long long)k * (m.vertCount - 1) / 63].position, 12);
        if (m.indices && m.indicesCount > 0)
            for (int k = 0; k < 64; k++) mix(h, &m.indices[(long long)k * (m.indicesCount - 1) / 63], 4);
    }
    return h ? h : 1ull;
}

}  // namespace

void rayTrace(int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config, curandState* /*randStates*/) {
    g_lastStatus = RT_OK;
    if (!g_defaultCtx) {
        const char* e = std::getenv("RT_DEVICE");
        int device = e ? std::atoi(e) : 0;
        g_lastStatus = rt_create(1, &device, &g_defaultCtx);
        if (g_lastStatus != RT_OK) { std::cerr << "rayTrace: " << rt_last_error() << std::endl; g_defaultCtx = nullptr; return; }
        g_uploadedObjects = g_uploadedMeshes = 0;
    }
    SceneDescStorage storage;
    rt_scene_desc desc;
    describeScene(scene, desc, storage);
    const unsigned long long objects = objectsFingerprint(desc, storage), meshes = meshesFingerprint(desc, storage);
    const SceneChange changed = (meshes != g_uploadedMeshes || g_uploadedObjects == 0) ? kSceneEverything
                              : (objects != g_uploadedObjects ? kSceneObjects : kSceneSame);
    g_lastStatus = rayTrace(g_defaultCtx, width, height, framebuffer, scene, config, changed, nullptr);
    if (g_lastStatus != RT_OK) { std::cerr << "rayTrace: " << rt_last_error() << std::endl; return; }
    g_uploadedObjects = objects; g_uploadedMeshes = meshes;
}

int rayTraceStatus() { return g_lastStatus; }

void setupCurand(curandState* /*randStates*/, unsigned long /*seed*/) {}

void rayTraceShutdown() {
    if (g_defaultCtx) rt_destroy(g_defaultCtx);
    g_defaultCtx = nullptr;
    g_uploadedObjects = g_uploadedMeshes = 0;
}

int prepareFrame(rt_ctx* ctx, const Scene& scene, const SceneConfig& config, SceneChange sceneChanged) {
    // the scene arrays live in HBM; they are re-flattened only when the host replaced them, and the meshes only when it was
    // not just the tables of spheres, planes and boxes that changed
    if (sceneChanged != kSceneSame) {
        SceneDescStorage storage;
        rt_scene_desc desc;
        describeScene(scene, desc, storage);
        int rc = sceneChanged == kSceneObjects ? rt_update_objects(ctx, &desc) : rt_upload_scene(ctx, &desc);
        if (rc != RT_OK) return rc;
    }
    rt_camera cam = scene.cam.toRt();
    rt_config cfg = toRtConfig(config);
    rt_dynamic dyn{};
    auto rv = [](const vec3& v) { return rt_vec3{v[0], v[1], v[2]}; };
    if (!scene.lights.empty()) { dyn.setLight = 1; dyn.lightPosition = rv(scene.lights[0].position); }
    if (!scene.spheres.empty()) { dyn.setSphere0 = 1; dyn.sphere0Position = rv(scene.spheres[0].position); dyn.sphere0Radius = scene.spheres[0].radius; }
    if (!scene.planes.empty()) { dyn.setPlane0 = 1; dyn.plane0AmbientColour = rv(scene.planes[0].mat.ambientColour); }
    return rt_update_frame_state(ctx, &cam, &cfg, &dyn);
}

int rayTrace(rt_ctx* ctx, int width, int height, GLubyte* framebuffer, const Scene& scene, const SceneConfig& config,
             SceneChange sceneChanged, rt_frame_stats* stats) {
    int rc = prepareFrame(ctx, scene, config, sceneChanged);
    if (rc != RT_OK) return rc;
    return rt_render_frame(ctx, width, height, framebuffer, stats);
}
