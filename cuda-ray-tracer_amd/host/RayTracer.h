// Host mirror of RayTracer / Scene / SceneConfig (RayTracer.h:17-101, RayTracer.cpp:58-417): owns the scene tables,
// the configuration and the framebuffer, and renders through the C ABI instead of launching a kernel itself.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "AABB.h"
#include "Camera.h"
#include "Model.h"
#include "Plane.h"
#include "Sphere.h"
#include "Triangle.h"
#include "rt_api.h"
#include "vec3.h"

typedef unsigned char GLubyte;   // the framebuffer element type of the reference (gl/glew.h)

// RayTracer.h:17-34.  Arrays are std::vectors; the count members mirror the reference's fields and are kept in
// step by RayTracer (scene.spheres[i], scene.sphereCount etc. read as in the reference).
// What changed in the scene arrays since the previous call: nothing; only the tables of spheres, planes and boxes or the light's
// colour (addSphere / addPlane / addAABB, RayTracer.cpp:360-403: the meshes on the GPU stay, rt_update_objects); or anything else
// (a scene switch, a model: everything is flattened and uploaded again, rt_upload_scene).
enum SceneChange { kSceneSame = 0, kSceneObjects = 1, kSceneEverything = 2 };

struct Scene {
    std::vector<Sphere> spheres;  int sphereCount = 0;
    std::vector<Plane> planes;    int planeCount = 0;
    std::vector<AABB> AABBs;      int AABBCount = 0;
    std::vector<Light> lights;    int lightCount = 1;
    std::vector<Model> models;    int modelCount = 0;
    Camera cam;
};

// RayTracer.h:36-70, same members and defaults.
struct SceneConfig {
    int fps = 0;

    float shadowBias = 0.01f;
    bool renderHardShadows = false;
    bool renderSoftShadows = false;
    int softShadowRadius = 5;
    int softShadowNum = 15;
    float shadowIntensity = 0.5f;

    bool areaLightSpecularEffect = false;
    bool reflections = false;
    int maxDepth = 2;
    bool reflectPlanes = false;
    float sphereReflectionStrength = 0.5f;
    float planeReflectionStrength = 0.5f;
    float AABBReflectionStrength = 0.5f;
    float modelReflectionStrength = 0.5f;

    bool renderAABBs = true;
    bool renderModels = true;

    bool ambientLighting = true;
    bool diffuseLighting = true;
    bool specularLighting = true;
    bool antiAliasing = false;
    bool boundingBox = true;
    bool eightBoundingBoxes = false;
    bool octree = false;
    bool cullBackTriangles = true;

    vec3 backgroundCol = vec3(0.1f, 0.1f, 0.1f);
    int backgroundBrightness = 7;
    int floorBrightness = 3;
};

rt_config toRtConfig(const SceneConfig& c);

// Backing arrays of an rt_scene_desc view of a Scene.
struct SceneDescStorage {
    std::vector<rt_sphere> spheres; std::vector<rt_plane> planes; std::vector<rt_aabb> boxes;
    std::vector<rt_light> lights; std::vector<rt_model> models;
};

// Fills `out` with a view of `scene` for rt_upload_scene (pointers into `storage` and into the scene's models,
// both of which the caller keeps alive until the upload returns).
void describeScene(const Scene& scene, rt_scene_desc& out, SceneDescStorage& storage);

class RayTracer {
public:
    Scene scene;
    SceneConfig config;

    std::vector<Scene> scenes;
    int sceneCount = 0;

    GLubyte* framebuffer = nullptr;   // width * height * 3 bytes, row-major RGB8 (host memory)
    int width = 0, height = 0;

    std::string modelDirectory = "models";   // where initialiseScenes looks for <name>.obj / .mtl
    int device = 0;
    // When set, launchKernel hands the prepared frame to this function instead of rt_render_frame: how a process that is one
    // rank of an image-tile sharded run renders (it returns an rt_status; framebuffer is only filled on rank 0).
    std::function<int(rt_ctx*, int, int, GLubyte*)> renderSharded;
    double lastGpuMs = 0.0;                   // device time of the last frame
    unsigned long long lastRayCount = 0;

    RayTracer() {}
    ~RayTracer();
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    void init(int width, int height);
    void initialiseScenes();
    void switchScene(int sceneNum);
    void launchKernel();

    void addSphere(vec3 pos, float radius, Material mat, ObjectType objectType = Diffuse);
    void addPlane(vec3 pos, vec3 n, Material mat, ObjectType objectType = Diffuse);
    void addAABB(vec3 pos, float size, Material mat, ObjectType objectType = Diffuse);

private:
    rt_ctx* ctx_ = nullptr;
    SceneChange sceneDirty_ = kSceneEverything;      // what of the scene arrays the next frame has to hand to the GPU
    std::vector<GLubyte> frameStorage_;
};
