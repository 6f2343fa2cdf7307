/*
 * ref_model_driver.cpp — runs the REFERENCE'S OWN Model::hit (Model.cpp:161-357) on arrays read from a file (test
 * infrastructure only).
 *
 * oracle/build_ref.sh compiles this file against the reference's headers and links it with the reference's Model.cpp,
 * utils.cpp and OBJ_Loader.cpp into the program oracle/_ref/ref_model_kat.  Model.cpp also refers to AABB members
 * (AABB.cpp does not compile with g++: AABB.cpp:90 binds a temporary to `vec3&`, which only MSVC accepts), to Octree and to
 * cudaMallocManaged; the program is linked with those references left unresolved and never reaches them:
 *   - the Model object lives in zeroed storage and no constructor runs (Model::Model would construct its AABB members);
 *   - only Model::hit's last branch is called (accelerate = accelerateTwice = useOctree = false, Model.cpp:318-356): the loop
 *     over all triangles with Model::triangleIntersect (Model.cpp:135-159, private, reached through hit), the closest-hit
 *     rule `t_tmp < tClosest` and the attribute interpolation `u*v0 + v*v1 + (1-u-v)*v2`.  That branch touches no box.
 * No arithmetic happens in this file.
 *
 * usage: ref_model_kat mesh|tri <input file> <output file>
 *   mesh  in : int32 {vertices, indices, rays, cull}; float[vertices][12] position, normal, ambient, diffuse;
 *              int32[indices]; float[rays][6] origin, direction
 *   tri   in : int32 {cases, cull, 0, 0}; float[cases][15] v0, v1, v2, origin, direction.  Each case is a model of its own
 *              whose index list names the triangle twice (the loop stops before the last triangle, Model.cpp:324), with
 *              vertex normals (1,0,0), (0,1,0), (0,0,1), so that the interpolated normal is (u, v, 1-u-v).
 *   out      : int32[n] hit; float[n] t0; float[n][9] interpolated normal, ambient, diffuse (as left by the last hit when hit = 0)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Model.h"

namespace {

template <typename T> bool readAll(FILE* f, T* p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }
template <typename T> bool writeAll(FILE* f, const T* p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }

Model* rawModel() {
    // zeroed storage instead of a constructed object: see the header
    return static_cast<Model*>(calloc(1, sizeof(Model)));
}

void setVertex(Vertex& v, const float* f12) {
    std::memset((void*)&v, 0, sizeof(Vertex));
    v.position = vec3(f12[0], f12[1], f12[2]);
    v.normal = vec3(f12[3], f12[4], f12[5]);
    v.ambient = vec3(f12[6], f12[7], f12[8]);
    v.diffuse = vec3(f12[9], f12[10], f12[11]);
}

void putHit(float* out9, const Vertex& hv) {
    const float f[9] = {hv.normal.x(), hv.normal.y(), hv.normal.z(), hv.ambient.x(), hv.ambient.y(), hv.ambient.z(),
                        hv.diffuse.x(), hv.diffuse.y(), hv.diffuse.z()};
    std::memcpy(out9, f, sizeof f);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s mesh|tri <in> <out>\n", argv[0]); return 2; }
    const bool meshMode = !strcmp(argv[1], "mesh");
    if (!meshMode && strcmp(argv[1], "tri")) { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
    FILE* in = fopen(argv[2], "rb");
    if (!in) { perror(argv[2]); return 1; }
    int header[4];
    if (!readAll(in, header, 4)) { fprintf(stderr, "short header\n"); return 1; }

    Model* model = rawModel();
    std::vector<int> hit;
    std::vector<float> t0, attrs;
    if (meshMode) {
        const int nVerts = header[0], nIndices = header[1], nRays = header[2];
        const bool cull = header[3] != 0;
        std::vector<float> vf((size_t)nVerts * 12), rays((size_t)nRays * 6);
        std::vector<int> indices((size_t)nIndices);
        if (!readAll(in, vf.data(), vf.size()) || !readAll(in, indices.data(), indices.size()) || !readAll(in, rays.data(), rays.size())) {
            fprintf(stderr, "short input\n"); return 1;
        }
        for (int i : indices) if (i < 0 || i >= nVerts) { fprintf(stderr, "index out of range\n"); return 1; }
        Vertex* vertices = static_cast<Vertex*>(calloc((size_t)nVerts + 1, sizeof(Vertex)));
        for (int i = 0; i < nVerts; i++) setVertex(vertices[i], &vf[(size_t)i * 12]);
        model->vertices = vertices; model->vertCount = nVerts;
        model->indices = indices.data(); model->indicesCount = nIndices;
        hit.assign((size_t)nRays, 0); t0.assign((size_t)nRays, 0.0f); attrs.assign((size_t)nRays * 9, 0.0f);
        for (int r = 0; r < nRays; r++) {
            const float* q = &rays[(size_t)r * 6];
            float t1 = 0; Vertex hv; std::memset((void*)&hv, 0, sizeof hv);
            hit[(size_t)r] = model->hit(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), t0[(size_t)r], t1, hv, PrimaryRay, false, false, false, cull) ? 1 : 0;
            putHit(&attrs[(size_t)r * 9], hv);
        }
    } else {
        const int nCases = header[0];
        const bool cull = header[1] != 0;
        std::vector<float> cases((size_t)nCases * 15);
        if (!readAll(in, cases.data(), cases.size())) { fprintf(stderr, "short input\n"); return 1; }
        Vertex* vertices = static_cast<Vertex*>(calloc(3, sizeof(Vertex)));
        int indices[6] = {0, 1, 2, 0, 1, 2};
        model->vertices = vertices; model->vertCount = 3;
        model->indices = indices; model->indicesCount = 6;
        vertices[0].normal = vec3(1, 0, 0); vertices[1].normal = vec3(0, 1, 0); vertices[2].normal = vec3(0, 0, 1);
        hit.assign((size_t)nCases, 0); t0.assign((size_t)nCases, 0.0f); attrs.assign((size_t)nCases * 9, 0.0f);
        for (int c = 0; c < nCases; c++) {
            const float* q = &cases[(size_t)c * 15];
            for (int k = 0; k < 3; k++) vertices[k].position = vec3(q[3 * k], q[3 * k + 1], q[3 * k + 2]);
            float t1 = 0; Vertex hv; std::memset((void*)&hv, 0, sizeof hv);
            hit[(size_t)c] = model->hit(vec3(q[9], q[10], q[11]), vec3(q[12], q[13], q[14]), t0[(size_t)c], t1, hv, PrimaryRay, false, false, false, cull) ? 1 : 0;
            putHit(&attrs[(size_t)c * 9], hv);
        }
    }
    fclose(in);
    FILE* out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 1; }
    const bool ok = writeAll(out, hit.data(), hit.size()) && writeAll(out, t0.data(), t0.size()) && writeAll(out, attrs.data(), attrs.size());
    if (fclose(out) != 0 || !ok) { fprintf(stderr, "write failed\n"); return 1; }
    return 0;
}
