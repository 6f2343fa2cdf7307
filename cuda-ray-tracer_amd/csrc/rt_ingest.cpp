// Host-only scene-ingest helpers of the C ABI: camera basis and model baking.  No GPU calls.
#include <cmath>
#include <cstring>
#include <string>

#include "rt_api.h"

namespace {

struct Box3 {
    float lo[3] = {0.0f, 0.0f, 0.0f};   // a fresh box is the single point (0,0,0): AABB.cpp:3-6
    float hi[3] = {0.0f, 0.0f, 0.0f};
    // AABB.cpp:26-34: a point outside moves the face 0.1 beyond it (the 0.1 is a double literal)
    void extendBy(const float p[3]) {
        for (int a = 0; a < 3; a++) {
            if (p[a] < lo[a]) lo[a] = (float)((double)p[a] - 0.1);
        }
        for (int a = 0; a < 3; a++) {
            if (p[a] > hi[a]) hi[a] = (float)((double)p[a] + 0.1);
        }
    }
};

inline float toRadians(float deg) { return (float)((double)deg * 0.01745329251994329576923690768489); }   // utils.h:102-104

void normalise3(float v[3]) {
    float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= len; v[1] /= len; v[2] /= len;
}

void cross3(const float a[3], const float b[3], float out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

}  // namespace

extern "C" {

int rt_camera_look(const float position[3], float fov, float yaw, float pitch, rt_camera* out) {
    if (!position || !out) return RT_ERR_INVALID;
    const float pitchR = toRadians(pitch), yawR = toRadians(yaw);
    float dir[3] = {std::cos(yawR) * std::cos(pitchR), std::sin(pitchR), std::sin(yawR) * std::cos(pitchR)};
    normalise3(dir);
    const float worldUp[3] = {0.0f, 1.0f, 0.0f};
    float right[3], up[3];
    cross3(dir, worldUp, right);
    normalise3(right);
    cross3(right, dir, up);
    normalise3(up);
    out->position = rt_vec3{position[0], position[1], position[2]};
    out->direction = rt_vec3{dir[0], dir[1], dir[2]};
    out->right = rt_vec3{right[0], right[1], right[2]};
    out->up = rt_vec3{up[0], up[1], up[2]};
    out->fovFactor = std::tan(toRadians(fov / 2));   // Camera.cpp:10
    return RT_OK;
}

int rt_bake_model(const float* pos, const float* nrm, const float* tex, const float* ka, const float* kd, const float* ks,
                  int vert_count, const uint32_t* loaded_indices, int indices_count, const float position[3], int size,
                  rt_vertex* vertices, int32_t* indices, float* bb, float* chunk_boxes, int32_t* chunk_indices,
                  int32_t* indices_per_box) {
    if (!pos || !nrm || !tex || !ka || !kd || !ks || !loaded_indices || !position || !vertices || !indices || !bb ||
        !chunk_boxes || !chunk_indices || !indices_per_box || vert_count < 0 || indices_count < 0)
        return RT_ERR_INVALID;
    for (int i = 0; i < indices_count; i++) {
        if (loaded_indices[i] >= (uint32_t)vert_count) return RT_ERR_INVALID;
    }
    if (indices_count > 0 && vert_count == 0) return RT_ERR_INVALID;

    // Model.cpp:62-63: triangles are dealt to the eight chunk boxes in file order, ceil(n / 8) per box
    const int perBoxTris = (int)std::ceil((double)(indices_count / 3) / 8.0);
    const int perBox = perBoxTris * 3;
    *indices_per_box = perBox;
    std::memset(indices, 0, sizeof(int32_t) * (size_t)indices_count);
    std::memset(chunk_indices, 0, sizeof(int32_t) * (size_t)perBox * 8);

    // Model.cpp:70-84: the loop stops one triangle short of the end
    int box = 0, slot = 0;
    for (int i = 0; i + 3 < indices_count; i += 3) {
        for (int k = 0; k < 3; k++) {
            const int32_t v = (int32_t)loaded_indices[i + k];
            indices[i + k] = v;
            chunk_indices[(size_t)box * perBox + slot + k] = v;
        }
        slot += 3;
        if (slot >= perBox) { box++; slot = 0; }
    }

    // Model.cpp:88-117: scale by the integer size, mirror Y, translate; grow the whole-model box vertex by vertex
    Box3 whole;
    const float s = (float)size;
    for (int j = 0; j < vert_count; j++) {
        rt_vertex v;
        std::memset(&v, 0, sizeof v);
        v.position = rt_vec3{position[0] + pos[3 * j] * s, position[1] + -pos[3 * j + 1] * s, position[2] + pos[3 * j + 2] * s};
        v.textureCoords = rt_vec3{tex[2 * j], tex[2 * j + 1], 0.0f};
        v.normal = rt_vec3{nrm[3 * j], -nrm[3 * j + 1], nrm[3 * j + 2]};
        v.ambient = rt_vec3{ka[3 * j], ka[3 * j + 1], ka[3 * j + 2]};
        v.diffuse = rt_vec3{kd[3 * j], kd[3 * j + 1], kd[3 * j + 2]};
        v.specular = rt_vec3{ks[3 * j], ks[3 * j + 1], ks[3 * j + 2]};
        const float p[3] = {v.position.x, v.position.y, v.position.z};
        whole.extendBy(p);
        vertices[j] = v;
    }
    std::memcpy(bb, whole.lo, sizeof whole.lo);
    std::memcpy(bb + 3, whole.hi, sizeof whole.hi);

    // Model.cpp:21-27: each chunk box grows over all of its index slots, unfilled ones (index 0) included
    for (int b = 0; b < 8; b++) {
        Box3 cb;
        for (int j = 0; j < perBox; j++) {
            const rt_vec3& q = vertices[chunk_indices[(size_t)b * perBox + j]].position;
            const float p[3] = {q.x, q.y, q.z};
            cb.extendBy(p);
        }
        std::memcpy(chunk_boxes + 6 * b, cb.lo, sizeof cb.lo);
        std::memcpy(chunk_boxes + 6 * b + 3, cb.hi, sizeof cb.hi);
    }
    return RT_OK;
}

}  // extern "C"
