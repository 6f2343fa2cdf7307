/*
 * rt_types.h — plain-C data types that cross the renderer's C ABI.
 *
 * Every struct is made of 32-bit floats, 32-bit ints and pointers only, so it
 * can be filled from C, C++, or a ctypes/cffi binding.  Field names follow the
 * reference's scene vocabulary; each type cites the reference declaration it
 * stands in for (paths relative to the reference checkout).
 *
 * Ownership: everything reachable from an rt_scene_desc is caller-owned and is
 * only read during rt_upload_scene(); the library keeps its own device copies.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vec3.h:6-35 — three packed floats. */
typedef struct rt_vec3 { float x, y, z; } rt_vec3;

/* utils.h:14-16 (ObjectName) */
enum rt_object_name {
    RT_SPHERE_T = 0, RT_PLANE_T = 1, RT_BOX_T = 2, RT_TRIANGLE_T = 3, RT_MODEL_T = 4, RT_AABB_T = 5
};
/* utils.h:18-21 (ObjectType) */
enum rt_object_type { RT_DIFFUSE = 0, RT_REFLECT = 1 };
/* utils.h:23-27 (RayType) */
enum rt_ray_type { RT_PRIMARY_RAY = 0, RT_SHADOW_RAY = 1, RT_REFLECT_RAY = 2 };

/* utils.h:39-46 (Material) */
typedef struct rt_material {
    rt_vec3 ambientColour;
    rt_vec3 diffuseColour;
    float ambient;
    float diffuse;
    float specular;
    float shininess;
} rt_material;

/* utils.h:29-37 (Light) — only position and colour are ever read on the render path. */
typedef struct rt_light {
    rt_vec3 position;
    rt_vec3 colour;
} rt_light;

/* Sphere.h:6-21 + Object.h:15-19 */
typedef struct rt_sphere {
    rt_vec3 position;
    float radius;
    rt_material mat;
    int32_t objectType;   /* rt_object_type */
    int32_t debug;        /* 0/1 */
} rt_sphere;

/* Plane.h:6-22 + Object.h:15-19 */
typedef struct rt_plane {
    rt_vec3 position;
    rt_vec3 n;
    rt_material mat;
    int32_t objectType;
    int32_t debug;
} rt_plane;

/* AABB.h:6-26 + Object.h:15-19 (renderable box) */
typedef struct rt_aabb {
    rt_vec3 min;
    rt_vec3 max;
    rt_material mat;
    int32_t objectType;
    int32_t debug;
} rt_aabb;

/* utils.h:59-67 (Vertex) — same 76-byte field order. */
typedef struct rt_vertex {
    rt_vec3 position;
    rt_vec3 textureCoords;
    rt_vec3 normal;
    int32_t materialIndex;
    rt_vec3 ambient;
    rt_vec3 diffuse;
    rt_vec3 specular;
} rt_vertex;

/* One of a model's eight sequential-chunk boxes: Model.h:32-33, AABB.h:22-25. */
typedef struct rt_chunk_box {
    rt_vec3 min;
    rt_vec3 max;
    int32_t indicesCount;
    const int32_t* includedModelIndices;   /* indicesCount entries */
} rt_chunk_box;

/*
 * Octree.h:6-17 + OctreeNode.h:6-20, redefined: the octree of one model as the flat arrays the kernels read instead of
 * pointer-linked nodes.  Produced by rt_build_octree; a host that keeps the reference's Octree class hands it back
 * through rt_model.octree and rt_upload_scene then uses it as is (after checking it) instead of building its own.
 *   nodes      nodeCount records of 128 bytes (one cache line): origin + power-of-two steps, { childBase, internal and
 *              leaf masks, model }, eight child boxes quantised to 8 bits per plane, eight leaf references.  Node 0 is
 *              the root; the internal children of a node are contiguous from childBase, in slot order.
 *   triangles  triangleRecords records of 48 bytes in leaf order: { v0, id word } { v1 - v0, model } { v2 - v0, 0 };
 *              the id word holds the triangle's index in the model and marks the last triangle of a leaf (bit 31).
 */
typedef struct rt_flat_octree {
    const void* nodes;        int64_t nodeCount;
    const void* triangles;    int64_t triangleRecords;
    int32_t leafCount;
    int32_t depth;            /* deepest nesting of internal nodes */
    int32_t maxDepth, leafTriangles;   /* the parameters it was built with (0 = library default) */
} rt_flat_octree;

/* Model.h:10-44 — a baked (world-space) mesh with its bounding volumes. */
typedef struct rt_model {
    rt_vec3 position;
    int32_t size;
    rt_material mat;
    int32_t objectType;
    int32_t debug;
    rt_vec3 bbMin;              /* boundingBox.min  (Model.h:30) */
    rt_vec3 bbMax;              /* boundingBox.max */
    rt_chunk_box boundingBoxes[8];   /* Model.h:33 */
    const rt_vertex* vertices;  /* vertCount entries (Model.h:36-37) */
    int32_t vertCount;
    const int32_t* indices;     /* indicesCount entries (Model.h:42-43) */
    int32_t indicesCount;
    const rt_flat_octree* octree;   /* Model.h:39 (Octree octree): prebuilt by rt_build_octree for exactly this model, or NULL */
} rt_model;

/* Camera.h:6-45 — the fields the render path reads (basis is recomputed on the host). */
typedef struct rt_camera {
    rt_vec3 position;
    rt_vec3 up;
    rt_vec3 right;
    rt_vec3 direction;
    float fovFactor;
} rt_camera;

/* RayTracer.h:36-70 (SceneConfig) — bools widened to int32. */
typedef struct rt_config {
    int32_t fps;
    float shadowBias;
    int32_t renderHardShadows;
    int32_t renderSoftShadows;
    int32_t softShadowRadius;
    int32_t softShadowNum;
    float shadowIntensity;
    int32_t areaLightSpecularEffect;
    int32_t reflections;
    int32_t maxDepth;
    int32_t reflectPlanes;
    float sphereReflectionStrength;
    float planeReflectionStrength;
    float AABBReflectionStrength;
    float modelReflectionStrength;
    int32_t renderAABBs;
    int32_t renderModels;
    int32_t ambientLighting;
    int32_t diffuseLighting;
    int32_t specularLighting;
    int32_t antiAliasing;
    int32_t boundingBox;
    int32_t eightBoundingBoxes;
    int32_t octree;
    int32_t cullBackTriangles;
    rt_vec3 backgroundCol;
    int32_t backgroundBrightness;
    int32_t floorBrightness;
} rt_config;

/* RayTracer.h:17-34 (Scene) without the camera, which travels per frame. */
typedef struct rt_scene_desc {
    const rt_sphere* spheres;  int32_t sphereCount;
    const rt_plane*  planes;   int32_t planeCount;
    const rt_aabb*   AABBs;    int32_t AABBCount;
    const rt_light*  lights;   int32_t lightCount;
    const rt_model*  models;   int32_t modelCount;
    /* Tree build parameters (0 = library default: depth 8; 2 triangles per leaf built on the host, 6 on the GPU). */
    int32_t octreeMaxDepth;
    int32_t octreeLeafTriangles;
    /* 0: the octrees are built on the host; 1: on the GPU (Morton sort, one launch per level) - same device layout,
     * different but equally valid tree shape, identical pixels.  Meant for uploads of millions of triangles. */
    int32_t buildOnDevice;
} rt_scene_desc;

/*
 * The scene members the reference's frame loop rewrites between launches
 * (Window.cpp:357-384): the light, its marker sphere spheres[0], and the
 * floor plane's ambient colour.  Each block is applied only if its flag is set.
 */
typedef struct rt_dynamic {
    int32_t setLight;       rt_vec3 lightPosition;
    int32_t setSphere0;     rt_vec3 sphere0Position;  float sphere0Radius;
    int32_t setPlane0;      rt_vec3 plane0AmbientColour;
} rt_dynamic;

/* Per-frame measurements returned by the render entry points. */
typedef struct rt_frame_stats {
    double   gpu_ms;          /* device time of the frame (HIP events on the render stream) */
    uint64_t rays;            /* (origin, direction) pairs traced this frame */
    uint64_t node_tests;      /* octree child boxes slab-tested (stats launches only) */
    uint64_t tri_tests;       /* ray/triangle tests (stats launches only) */
    uint64_t attr_fetches;    /* closest-hit vertex attribute fetches (stats launches only) */
    uint64_t gate_tests;      /* whole-model bounding-box gate tests (stats launches only) */
    uint64_t split_waves;     /* extra waves the most recent block selection chose for the frame after it (0 = nothing split) */
    uint64_t node_visits;     /* octree nodes fetched, one 128-byte record each (stats launches only) */
    uint64_t table_flops;     /* flops of the sphere / plane / box tests made on the linear tables: 30 / 14 / 26 each (stats launches only) */
    /* The traversal works on several pending entries of a ray at once, so it also enters subtrees that the hit it finds later
     * would have pruned.  These count what a traversal that knew every ray's hit distance in advance would do: for closest-hit
     * traces a second, counting-only pass bounded by the result (rt_render_frame_ids and rt_count_rays run it); for any-hit
     * traces what they did.  node_tests - useful_node_tests etc. is the speculative share. */
    uint64_t useful_node_tests;
    uint64_t useful_tri_tests;
    uint64_t useful_node_visits;
} rt_frame_stats;

/* Sizes of the flattened device scene, for roofline accounting and tests. */
typedef struct rt_scene_info {
    int32_t sphereCount, planeCount, AABBCount, modelCount;
    int64_t triangleCount;     /* triangles the render path can hit (last one per model excluded) */
    int64_t nodeCount;         /* 128-byte octree nodes */
    int64_t leafCount;
    int32_t maxDepth;          /* deepest level reached */
    int64_t deviceBytes;       /* bytes resident in HBM for the scene */
} rt_scene_info;

#ifdef __cplusplus
}
#endif
#endif /* RT_TYPES_H */
