// Small C shim over the host mirror so that tests can drive RayTracer / Model / objl::Loader through ctypes.
#include <cstring>
#include <string>

#include "Headless.h"
#include "OBJ_Loader.h"
#include "Octree.h"
#include "kernels.h"

extern "C" {

void* rth_create(int width, int height, const char* modelDirectory, int device) { return new Headless(width, height, modelDirectory ? modelDirectory : "", device); }
void rth_destroy(void* h) { delete static_cast<Headless*>(h); }
int rth_key_preset(void* h, char key) { return static_cast<Headless*>(h)->keyPreset(key) ? 1 : 0; }
void rth_key_held(void* h, char key) { static_cast<Headless*>(h)->keyHeld(key); }
void rth_display(void* h) { static_cast<Headless*>(h)->display(); }
const unsigned char* rth_framebuffer(void* h) { return static_cast<Headless*>(h)->rayTracer.framebuffer; }
unsigned long long rth_last_rays(void* h) { return static_cast<Headless*>(h)->rayTracer.lastRayCount; }
// Renders the handle's current scene through the entry point with the reference's own argument list (kernels.h:11) into
// `out` (width * height * 3 bytes); returns its rt_status.  The per-frame fix-ups must have been applied (rth_display).
int rth_raytrace_reference(void* h, unsigned char* out) {
    Headless* app = static_cast<Headless*>(h);
    setupCurand(nullptr, 0ul);
    rayTrace(app->rayTracer.width, app->rayTracer.height, out, app->rayTracer.scene, app->rayTracer.config, nullptr);
    return rayTraceStatus();
}
void rth_raytrace_shutdown() { rayTraceShutdown(); }
void rth_add_sphere(void* h, const float* pos, float radius, const rt_material* m, int type) {
    Material mm{vec3(m->ambientColour.x, m->ambientColour.y, m->ambientColour.z), vec3(m->diffuseColour.x, m->diffuseColour.y, m->diffuseColour.z),
                m->ambient, m->diffuse, m->specular, m->shininess};
    static_cast<Headless*>(h)->rayTracer.addSphere(vec3(pos[0], pos[1], pos[2]), radius, mm, (ObjectType)type);
}

// Current scene / camera / config as C-ABI structs.  The returned descriptor points into storage owned by the
// handle and stays valid until the next rth_describe call on it.
struct DescHolder { SceneDescStorage storage; rt_scene_desc desc; };
static thread_local DescHolder g_holder;
const rt_scene_desc* rth_describe(void* h) {
    Headless* app = static_cast<Headless*>(h);
    // apply the per-frame fix-ups first, exactly as display() would before rendering
    Scene& s = app->rayTracer.scene; SceneConfig& c = app->rayTracer.config;
    if (!s.spheres.empty()) s.spheres[0].radius = (float)c.softShadowRadius;
    if (!s.planes.empty()) s.planes[0].mat.ambientColour = vec3(0.1f, 0.1f, 0.1f) * (float)c.floorBrightness;
    c.backgroundCol = vec3(0.1f, 0.1f, 0.1f) * (float)c.backgroundBrightness;
    describeScene(s, g_holder.desc, g_holder.storage);
    return &g_holder.desc;
}
void rth_camera(void* h, rt_camera* out) { *out = static_cast<Headless*>(h)->rayTracer.scene.cam.toRt(); }
void rth_config(void* h, rt_config* out) { *out = toRtConfig(static_cast<Headless*>(h)->rayTracer.config); }

// Model / loader
void* rth_model_load(const float* pos, int size, const char* filename, const rt_material* m, int debug, int type) {
    Material mm{vec3(m->ambientColour.x, m->ambientColour.y, m->ambientColour.z), vec3(m->diffuseColour.x, m->diffuseColour.y, m->diffuseColour.z),
                m->ambient, m->diffuse, m->specular, m->shininess};
    return new Model(vec3(pos[0], pos[1], pos[2]), size, filename, mm, debug != 0, (ObjectType)type);
}
void rth_model_free(void* m) { delete static_cast<Model*>(m); }
int rth_model_loaded(void* m) { return static_cast<Model*>(m)->loaded ? 1 : 0; }
void rth_model_view(void* m, rt_model* out) { *out = static_cast<Model*>(m)->toRt(); }
void rth_model_counts(void* m, int* out4) {
    Model* mm = static_cast<Model*>(m);
    out4[0] = mm->vertCount; out4[1] = mm->indicesCount; out4[2] = mm->materialCount; out4[3] = mm->meshCount;
}
// Builds the host pointer octree over the model's reachable triangles; returns node count, leaf count (non-empty), and
// the number of triangles found in the leaves.
void rth_model_octree_stats(void* m, int maxDepth, int leafTriangles, int* out3) {
    Model* mm = static_cast<Model*>(m);
    Octree tree; tree.maxDepth = maxDepth; tree.leafTriangles = leafTriangles;
    std::vector<Vertex> soup;
    mm->buildOctree(tree, soup);
    struct Walk { static int tris(const OctreeNode* n) { if (n->isLeaf) return n->verticesCount / 3; int t = 0; for (int k = 0; k < 8; k++) t += tris(&n->children[k]); return t; } };
    out3[0] = tree.nodeCount(); out3[1] = tree.leafCount(); out3[2] = Walk::tris(tree.root);
}

// Loader only: objl::Loader::LoadFile on `path`; the handle gives access to everything the loader produced.
struct LoaderHolder { objl::Loader loader; bool ok = false; };
void* rth_loader_load(const char* path) {
    LoaderHolder* h = new LoaderHolder();
    h->ok = h->loader.LoadFile(path);
    return h;
}
void rth_loader_free(void* h) { delete static_cast<LoaderHolder*>(h); }
// out5: LoadFile's return value, LoadedVertices, LoadedIndices, LoadedMeshes, LoadedMaterials
void rth_loader_counts(void* h, int* out5) {
    LoaderHolder* l = static_cast<LoaderHolder*>(h);
    out5[0] = l->ok ? 1 : 0; out5[1] = (int)l->loader.LoadedVertices.size(); out5[2] = (int)l->loader.LoadedIndices.size();
    out5[3] = (int)l->loader.LoadedMeshes.size(); out5[4] = (int)l->loader.LoadedMaterials.size();
}
static void dumpVertices(const std::vector<objl::Vertex>& v, float* out8) {
    for (size_t i = 0; i < v.size(); i++) {
        const float f[8] = {v[i].Position.X, v[i].Position.Y, v[i].Position.Z, v[i].Normal.X, v[i].Normal.Y, v[i].Normal.Z,
                            v[i].TextureCoordinate.X, v[i].TextureCoordinate.Y};
        std::memcpy(out8 + 8 * i, f, sizeof f);
    }
}
// LoadedVertices as 8 floats each (position, normal, texture coordinate) and LoadedIndices
void rth_loader_arrays(void* h, float* vertices8, unsigned int* indices) {
    LoaderHolder* l = static_cast<LoaderHolder*>(h);
    dumpVertices(l->loader.LoadedVertices, vertices8);
    for (size_t i = 0; i < l->loader.LoadedIndices.size(); i++) indices[i] = l->loader.LoadedIndices[i];
}
// Mesh `m`: counts2 = (vertices, indices); name and material name copied into buffers of `cap` bytes; material13 = Ka, Kd,
// Ks, Ns, Ni, d, illum; vertices8 may be null, else room for the mesh's vertices
void rth_loader_mesh(void* h, int m, int* counts2, char* name, char* materialName, int cap, float* material13, float* vertices8) {
    const objl::Mesh& mesh = static_cast<LoaderHolder*>(h)->loader.LoadedMeshes[(size_t)m];
    counts2[0] = (int)mesh.Vertices.size(); counts2[1] = (int)mesh.Indices.size();
    std::strncpy(name, mesh.MeshName.c_str(), (size_t)cap - 1); name[cap - 1] = 0;
    const objl::Material& mt = mesh.MeshMaterial;
    std::strncpy(materialName, mt.name.c_str(), (size_t)cap - 1); materialName[cap - 1] = 0;
    const float f[13] = {mt.Ka.X, mt.Ka.Y, mt.Ka.Z, mt.Kd.X, mt.Kd.Y, mt.Kd.Z, mt.Ks.X, mt.Ks.Y, mt.Ks.Z, mt.Ns, mt.Ni, mt.d, (float)mt.illum};
    std::memcpy(material13, f, sizeof f);
    if (vertices8) dumpVertices(mesh.Vertices, vertices8);
}
// Material `k` of LoadedMaterials: name, map_Kd, and the 13 numbers as above
void rth_loader_material(void* h, int k, char* name, char* mapKd, int cap, float* material13) {
    const objl::Material& mt = static_cast<LoaderHolder*>(h)->loader.LoadedMaterials[(size_t)k];
    std::strncpy(name, mt.name.c_str(), (size_t)cap - 1); name[cap - 1] = 0;
    std::strncpy(mapKd, mt.map_Kd.c_str(), (size_t)cap - 1); mapKd[cap - 1] = 0;
    const float f[13] = {mt.Ka.X, mt.Ka.Y, mt.Ka.Z, mt.Kd.X, mt.Kd.Y, mt.Kd.Z, mt.Ks.X, mt.Ks.Y, mt.Ks.Z, mt.Ns, mt.Ni, mt.d, (float)mt.illum};
    std::memcpy(material13, f, sizeof f);
}

// Loader only: triangulation of one polygon given as n positions (3 floats each); returns the index count.
int rth_triangulate(const float* positions, int n, unsigned int* out, int cap) {
    std::vector<objl::Vertex> corners((size_t)n);
    for (int i = 0; i < n; i++) corners[(size_t)i].Position = objl::Vector3(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    std::vector<unsigned int> idx;
    objl::TriangulateFace(corners, idx);
    for (size_t i = 0; i < idx.size() && (int)i < cap; i++) out[i] = idx[i];
    return (int)idx.size();
}

}  // extern "C"
