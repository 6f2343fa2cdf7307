#include "RayTracer.h"

#include <iostream>

#include "kernels.h"

namespace {

rt_vec3 rv(const vec3& v) { return rt_vec3{v[0], v[1], v[2]}; }

rt_material rm(const Material& m) {
    rt_material o;
    o.ambientColour = rv(m.ambientColour); o.diffuseColour = rv(m.diffuseColour);
    o.ambient = m.ambient; o.diffuse = m.diffuse; o.specular = m.specular; o.shininess = m.shininess;
    return o;
}

Material mat(vec3 ka, vec3 kd, float a, float d, float s, float sh) { return Material{ka, kd, a, d, s, sh}; }

// The pieces every scene of the reference shares (RayTracer.cpp:79-93 and their repeats).
Sphere lightMarker(vec3 at) { return Sphere(at, 2.0f, mat(vec3(1, 1, 1), vec3(1, 1, 1), 1.0f, 0.0f, 0.0f, 200.0f), true); }
Plane floorPlane() { return Plane(vec3(0, 20, 0), vec3(0, 1, 0), mat(vec3(0.1f, 0.1f, 0.1f), vec3(0.1f, 0.1f, 0.1f), 0.5f, 0.5f, 0.0f, 200.0f), false, Reflect); }
Light pointLight(vec3 at) { return Light{at, vec3(1, 1, 1), vec3(0.5f, 0, 0), 4, vec3(0, 0, 0.5f), 2, 8}; }

void finish(Scene& s) {
    s.sphereCount = (int)s.spheres.size(); s.planeCount = (int)s.planes.size(); s.AABBCount = (int)s.AABBs.size();
    s.modelCount = (int)s.models.size(); s.lightCount = (int)s.lights.size();
}

}  // namespace

rt_config toRtConfig(const SceneConfig& c) {
    rt_config o;
    o.fps = c.fps; o.shadowBias = c.shadowBias; o.renderHardShadows = c.renderHardShadows; o.renderSoftShadows = c.renderSoftShadows;
    o.softShadowRadius = c.softShadowRadius; o.softShadowNum = c.softShadowNum; o.shadowIntensity = c.shadowIntensity;
    o.areaLightSpecularEffect = c.areaLightSpecularEffect; o.reflections = c.reflections; o.maxDepth = c.maxDepth;
    o.reflectPlanes = c.reflectPlanes; o.sphereReflectionStrength = c.sphereReflectionStrength;
    o.planeReflectionStrength = c.planeReflectionStrength; o.AABBReflectionStrength = c.AABBReflectionStrength;
    o.modelReflectionStrength = c.modelReflectionStrength; o.renderAABBs = c.renderAABBs; o.renderModels = c.renderModels;
    o.ambientLighting = c.ambientLighting; o.diffuseLighting = c.diffuseLighting; o.specularLighting = c.specularLighting;
    o.antiAliasing = c.antiAliasing; o.boundingBox = c.boundingBox; o.eightBoundingBoxes = c.eightBoundingBoxes; o.octree = c.octree;
    o.cullBackTriangles = c.cullBackTriangles; o.backgroundCol = rv(c.backgroundCol);
    o.backgroundBrightness = c.backgroundBrightness; o.floorBrightness = c.floorBrightness;
    return o;
}

RayTracer::~RayTracer() {
    if (ctx_) rt_destroy(ctx_);
}

void RayTracer::init(int width, int height) {
    this->width = width;
    this->height = height;
    frameStorage_.assign((size_t)width * (size_t)height * 3, 0);
    framebuffer = frameStorage_.data();
    if (!ctx_ && rt_create(1, &device, &ctx_) != RT_OK) {
        std::cerr << "HIP Error: " << rt_last_error() << " at " << __FILE__ << ", line: " << __LINE__ << std::endl;
        ctx_ = nullptr;
    }
    sceneCount = 9;
    initialiseScenes();
}

// The nine scenes of RayTracer.cpp:71-337, written as data.
void RayTracer::initialiseScenes() {
    scenes.clear();
    scenes.resize((size_t)sceneCount);
    const float aspect = width / (float)height;
    const std::string dir = modelDirectory.empty() ? std::string() : modelDirectory + "/";
    auto path = [&](const char* name) { return dir + name + ".obj"; };
    const Material tableMat = mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.1f, 0.9f, 0.0f, 10.0f);
    const Material chaliceMat = mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.1f, 0.9f, 1.0f, 1000.0f);
    auto tableSet = [&](Scene& s) {
        s.models.emplace_back(vec3(0, 19, 0), 15, path("Table").c_str(), tableMat, false, Diffuse);
        s.models.emplace_back(vec3(0, 6, 10), 10, path("Chalice").c_str(), chaliceMat, false, Reflect);
        s.models.emplace_back(vec3(0, 6, 0), 10, path("Book2_Open").c_str(), tableMat, false, Diffuse);
    };

    {   // 0: empty
        Scene& s = scenes[0];
        s.cam = Camera(vec3(0, 0, 50), 90.0f, aspect);
        s.spheres = {lightMarker(vec3(0, 0, -50))};
        s.lights = {pointLight(vec3(0, -40, -50))};
    }
    {   // 1: spheres for reflections
        Scene& s = scenes[1];
        s.cam = Camera(vec3(0, 0, 0), 90.0f, aspect);
        s.spheres = {lightMarker(vec3(-30, -70, 0)),
                     Sphere(vec3(0, 0, -70), 15.0f, mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.2f, 0.2f, 0.0f, 200.0f), false, Reflect),
                     Sphere(vec3(50, 0, -70), 15.0f, mat(vec3(0, 1, 0), vec3(0, 1, 0), 0.2f, 0.2f, 0.0f, 200.0f), false, Reflect),
                     Sphere(vec3(0, -100, -240), 100.0f, mat(vec3(0, 1, 1), vec3(0, 1, 0), 0.2f, 0.2f, 0.0f, 200.0f), false, Reflect)};
        s.AABBs = {AABB(vec3(-50, 0, -70), 15.0f, mat(vec3(1, 0, 1), vec3(1, 0, 1), 0.2f, 0.2f, 0.0f, 200.0f), false, Reflect)};
        s.lights = {pointLight(vec3(-30, -70, 0))};
    }
    {   // 2: 100 spheres
        Scene& s = scenes[2];
        s.cam = Camera(vec3(-50, 0, 0), 90.0f, aspect);
        s.spheres = {lightMarker(vec3(0, 0, -50))};
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++)
                s.spheres.push_back(Sphere(vec3((float)(j * 15.0), 0.0f, (float)(i * 15)), 2.0f, mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.1f, 0.9f, 0.9f, 200.0f), false));
        s.lights = {pointLight(vec3(0, -40, -50))};
    }
    {   // 3: boxes
        Scene& s = scenes[3];
        s.cam = Camera(vec3(1.75159f, -55.2005f, 144.58f), 90.0f, aspect, -93.5f, 6.6f);
        s.spheres = {lightMarker(vec3(-30, -70, 0))};
        const Material blue = mat(vec3(0, 0, 1), vec3(0, 0, 1), 0.2f, 0.2f, 0.3f, 10), yellow = mat(vec3(1, 1, 0), vec3(1, 1, 0), 0.2f, 0.2f, 0.3f, 10);
        const Material cyan = mat(vec3(0, 1, 1), vec3(0, 1, 1), 0.2f, 0.2f, 0.3f, 10);
        struct Row { float x, y, z, size; const Material* m; ObjectType t; };
        const Material red = mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.2f, 0.8f, 0.9f, 200), green = mat(vec3(0, 1, 0), vec3(0, 1, 0), 0.2f, 0.8f, 0.9f, 200);
        const Row rows[12] = {{100, 0, 0, 100, &blue, Reflect}, {-100, 0, 0, 100, &yellow, Reflect}, {0, 0, -100, 100, &cyan, Diffuse},
                              {100, -100, 0, 100, &blue, Reflect}, {-100, -100, 0, 100, &yellow, Reflect}, {0, -100, -100, 100, &cyan, Diffuse},
                              {0, 0, 10, 40, &red, Diffuse}, {0, -30, 10, 20, &green, Diffuse}, {100, 0, 100, 100, &blue, Reflect},
                              {100, -100, 100, 100, &blue, Reflect}, {-100, 0, 100, 100, &yellow, Reflect}, {-100, -100, 100, 100, &yellow, Reflect}};
        for (const Row& r : rows) s.AABBs.push_back(AABB(vec3(r.x, r.y, r.z), r.size, *r.m, false, r.t));
        s.lights = {pointLight(vec3(-30, -70, 0))};
    }
    {   // 4: table + book + cup
        Scene& s = scenes[4];
        s.cam = Camera(vec3(-11.2339f, -4.81805f, -13.431f), 90.0f, aspect, -311.2f, 13.8f);
        s.spheres = {lightMarker(vec3(-30, -70, 0))};
        tableSet(s);
        s.lights = {pointLight(vec3(-30, -70, 0))};
    }
    {   // 5: spheres for shadows
        Scene& s = scenes[5];
        s.cam = Camera(vec3(0, 0, 0), 90.0f, aspect);
        s.spheres = {lightMarker(vec3(-29, -32, 0)),
                     Sphere(vec3(0, 0, -70), 15.0f, mat(vec3(1, 0, 1), vec3(1, 0, 1), 0.2f, 0.8f, 0.8f, 200.0f), false),
                     Sphere(vec3(20, 0, -120), 15.0f, mat(vec3(0, 1, 1), vec3(0, 1, 1), 0.2f, 0.8f, 0.8f, 200.0f), false),
                     Sphere(vec3(-20, 0, -120), 15.0f, mat(vec3(1, 1, 0), vec3(1, 1, 0), 0.2f, 0.8f, 0.8f, 200.0f), false)};
        s.lights = {pointLight(vec3(-29, -32, 0))};
    }
    {   // 6: single chest model
        Scene& s = scenes[6];
        s.cam = Camera(vec3(48.0561f, -33.9654f, 32.7296f), 90.0f, aspect, -146.2f, 19.7f);
        s.spheres = {lightMarker(vec3(21, -13, -18))};
        s.models.emplace_back(vec3(0, 0, 0), 15, path("Chest_gold").c_str(), mat(vec3(1, 0, 0), vec3(1, 0, 0), 0.1f, 0.9f, 0.9f, 200.0f), false, Diffuse);
        s.lights = {pointLight(vec3(21, -13, -18))};
    }
    {   // 7: reflective boxes + spheres
        Scene& s = scenes[7];
        s.cam = Camera(vec3(0.774152f, -77.2231f, 82.8138f), 90.0f, aspect, -89.5f, 28.7f);
        const Material pink = mat(vec3(1, 0, 1), vec3(1, 0, 1), 0.2f, 0.8f, 0.8f, 200.0f);
        const Material grey = mat(vec3(0.2f, 0.2f, 0.2f), vec3(0.2f, 0.2f, 0.2f), 0.5f, 0.2f, 0.3f, 10);
        s.spheres = {lightMarker(vec3(2, -73, -16)), Sphere(vec3(0, 0, 0), 15.0f, pink, false, Reflect),
                     Sphere(vec3(0, 0, -40), 15.0f, pink, false, Reflect), Sphere(vec3(0, 0, 40), 15.0f, pink, false, Reflect)};
        s.AABBs = {AABB(vec3(-100, 0, 0), 100.0f, grey, false, Reflect), AABB(vec3(100, 0, 0), 100.0f, grey, false, Reflect)};
        s.lights = {pointLight(vec3(2, -73, -16))};
    }
    {   // 8: reflective box + models
        Scene& s = scenes[8];
        s.cam = Camera(vec3(38.9299f, -39.5555f, 76.1236f), 90.0f, aspect, -129.6f, 15.0f);
        s.spheres = {lightMarker(vec3(2, -73, -16))};
        s.AABBs = {AABB(vec3(-100, 0, 0), 100.0f, mat(vec3(0.2f, 0.2f, 0.2f), vec3(0.2f, 0.2f, 0.2f), 0.5f, 0.2f, 0.3f, 10), false, Reflect)};
        tableSet(s);
        s.lights = {pointLight(vec3(2, -73, -16))};
    }
    for (Scene& s : scenes) {
        s.planes = {floorPlane()};
        finish(s);
    }
    scene = scenes[0];
    sceneDirty_ = kSceneEverything;
}

void RayTracer::switchScene(int sceneNum) {
    if (sceneNum < 0 || sceneNum >= sceneCount) return;
    scene = scenes[(size_t)sceneNum];
    sceneDirty_ = kSceneEverything;
}

void RayTracer::addSphere(vec3 pos, float radius, Material m, ObjectType objectType) {
    scene.spheres.push_back(Sphere(pos, radius, m, false, objectType));
    scene.sphereCount = (int)scene.spheres.size();
    if (sceneDirty_ == kSceneSame) sceneDirty_ = kSceneObjects;      // the meshes on the GPU stay
}

void RayTracer::addPlane(vec3 pos, vec3 n, Material m, ObjectType objectType) {
    scene.planes.push_back(Plane(pos, n, m, false, objectType));
    scene.planeCount = (int)scene.planes.size();
    if (sceneDirty_ == kSceneSame) sceneDirty_ = kSceneObjects;      // the meshes on the GPU stay
}

void RayTracer::addAABB(vec3 pos, float size, Material m, ObjectType objectType) {
    scene.AABBs.push_back(AABB(pos, size, m, false, objectType));
    scene.AABBCount = (int)scene.AABBs.size();
    if (sceneDirty_ == kSceneSame) sceneDirty_ = kSceneObjects;      // the meshes on the GPU stay
}

void describeScene(const Scene& scene, rt_scene_desc& out, SceneDescStorage& st) {
    st.spheres.clear(); st.planes.clear(); st.boxes.clear(); st.lights.clear(); st.models.clear();
    for (const Sphere& s : scene.spheres) st.spheres.push_back(rt_sphere{rv(s.position), s.radius, rm(s.mat), s.objectType, s.debug ? 1 : 0});
    for (const Plane& p : scene.planes) st.planes.push_back(rt_plane{rv(p.position), rv(p.n), rm(p.mat), p.objectType, p.debug ? 1 : 0});
    for (const AABB& b : scene.AABBs) st.boxes.push_back(rt_aabb{rv(b.min), rv(b.max), rm(b.mat), b.objectType, b.debug ? 1 : 0});
    for (const Light& l : scene.lights) st.lights.push_back(rt_light{rv(l.position), rv(l.colour)});
    for (const Model& m : scene.models) st.models.push_back(m.toRt());
    out = rt_scene_desc{};
    out.spheres = st.spheres.data(); out.sphereCount = (int32_t)st.spheres.size();
    out.planes = st.planes.data(); out.planeCount = (int32_t)st.planes.size();
    out.AABBs = st.boxes.data(); out.AABBCount = (int32_t)st.boxes.size();
    out.lights = st.lights.data(); out.lightCount = (int32_t)st.lights.size();
    out.models = st.models.data(); out.modelCount = (int32_t)st.models.size();
}

// RayTracer.cpp:405-417: render one frame, report a device error on stderr and carry on.
void RayTracer::launchKernel() {
    if (!ctx_) {
        std::cerr << "HIP Error: no render context at " << __FILE__ << ", line: " << __LINE__ << std::endl;
        return;
    }
    rt_frame_stats stats{};
    int rc;
    if (renderSharded) {
        rc = prepareFrame(ctx_, scene, config, sceneDirty_);
        if (rc == RT_OK) rc = renderSharded(ctx_, width, height, framebuffer);
    } else {
        rc = rayTrace(ctx_, width, height, framebuffer, scene, config, sceneDirty_, &stats);
    }
    if (rc != RT_OK) {
        std::cerr << "HIP Error: " << rt_last_error() << " at " << __FILE__ << ", line: " << __LINE__ << std::endl;
        return;
    }
    sceneDirty_ = kSceneSame;
    lastGpuMs = stats.gpu_ms;
    lastRayCount = stats.rays;
}
