/*
 * ref_driver.cpp — C entry points over the REFERENCE'S OWN compiled sources (test infrastructure only).
 *
 * This file is compiled by oracle/build_ref.sh against the reference's headers, from the checkout where it lies, and
 * linked with the reference's utils.cpp, Sphere.cpp, Plane.cpp, Triangle.cpp, Camera.cpp and OBJ_Loader.cpp into
 * oracle/_ref/libref.so.  It contains no arithmetic of its own: every number it returns was computed by the reference's
 * functions (solveQuadratic utils.cpp:10-32, Sphere::hit Sphere.cpp:17-44, Plane::hit Plane.cpp:16-22, Triangle::hit
 * Triangle.cpp:20-42, Camera::Camera / rasterToCameraSpace / updateDirection / move / mouseMovement Camera.cpp:5-83, the
 * inline vector operations of vec3.h:62-117, objl::Loader::LoadFile OBJ_Loader.cpp:183-465).  The loops below only feed
 * arrays through them.  Used by tests/ and by tests/golden/make_ref_golden.py to pin the CPU oracle, the host loader and
 * the HIP kernels against the reference itself; nothing in the product may load it.
 */
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "Camera.h"
#include "OBJ_Loader.h"
#include "Plane.h"
#include "Sphere.h"
#include "Triangle.h"
#include "utils.h"
#include "vec3.h"

namespace {

vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }
void put(float* out, const vec3& v) { out[0] = v.x(); out[1] = v.y(); out[2] = v.z(); }
const Material kNoMaterial = {vec3(), vec3(), 0, 0, 0, 0};

// The reference passes Camera by value to its kernel (kernels.h:11); tests receive the same bytes.
static_assert(sizeof(Camera) == 84, "Camera layout: up, right, direction, position, fov, fovFactor, aspectRatio, near, far, moveSpeed, sensitivity, pitch, yaw");

Camera cameraFromBytes(const unsigned char* bytes) {
    Camera cam;
    std::memcpy((void*)&cam, bytes, sizeof(Camera));
    return cam;
}

// The loader reports its progress on std::cout (OBJ_Loader.h: OBJL_CONSOLE_OUTPUT); keep that out of the test output.
struct QuietCout {
    std::ostringstream sink;
    std::streambuf* saved;
    QuietCout() : saved(std::cout.rdbuf(sink.rdbuf())) {}
    ~QuietCout() { std::cout.rdbuf(saved); }
};

struct LoaderHolder {
    objl::Loader loader;
    bool ok = false;
};

void dumpVertices(const std::vector<objl::Vertex>& v, float* out8) {
    for (size_t i = 0; i < v.size(); i++) {
        const float f[8] = {v[i].Position.X, v[i].Position.Y, v[i].Position.Z, v[i].Normal.X, v[i].Normal.Y, v[i].Normal.Z,
                            v[i].TextureCoordinate.X, v[i].TextureCoordinate.Y};
        std::memcpy(out8 + 8 * i, f, sizeof f);
    }
}

void dumpMaterial(const objl::Material& mt, float* out13) {
    const float f[13] = {mt.Ka.X, mt.Ka.Y, mt.Ka.Z, mt.Kd.X, mt.Kd.Y, mt.Kd.Z, mt.Ks.X, mt.Ks.Y, mt.Ks.Z, mt.Ns, mt.Ni, mt.d, (float)mt.illum};
    std::memcpy(out13, f, sizeof f);
}

void copyName(char* out, int cap, const std::string& s) {
    std::strncpy(out, s.c_str(), (size_t)cap - 1);
    out[cap - 1] = 0;
}

}  // namespace

extern "C" {

// sizeof of the reference's types, in the order: vec3, Material, Sphere, Plane, Triangle, Camera, Vertex, Light, Hit
void ref_sizes(int* out9) {
    const int s[9] = {(int)sizeof(vec3), (int)sizeof(Material), (int)sizeof(Sphere), (int)sizeof(Plane), (int)sizeof(Triangle),
                      (int)sizeof(Camera), (int)sizeof(Vertex), (int)sizeof(Light), (int)sizeof(Hit)};
    std::memcpy(out9, s, sizeof s);
}

// solveQuadratic, utils.cpp:10-32.  x0 / x1 keep their incoming values where the routine does not write them.
void ref_solve_quadratic(int n, const float* a, const float* b, const float* c, int* ok, float* x0, float* x1) {
    for (int i = 0; i < n; i++) ok[i] = solveQuadratic(a[i], b[i], c[i], x0[i], x1[i]) ? 1 : 0;
}

// Sphere::hit, Sphere.cpp:17-44
void ref_sphere_hit(int n, const float* centre, const float* radius, const float* origin, const float* dir, int* ok, float* t0, float* t1) {
    for (int i = 0; i < n; i++) {
        Sphere s(v3(centre + 3 * i), radius[i], kNoMaterial);
        ok[i] = s.hit(v3(origin + 3 * i), v3(dir + 3 * i), t0[i], t1[i]) ? 1 : 0;
    }
}

// Sphere::normalAt, Sphere.cpp:46-48
void ref_sphere_normal_at(int n, const float* centre, const float* point, float* out) {
    for (int i = 0; i < n; i++) {
        Sphere s(v3(centre + 3 * i), 1.0f, kNoMaterial);
        put(out + 3 * i, s.normalAt(v3(point + 3 * i)));
    }
}

// Plane::hit, Plane.cpp:16-22, and Plane::normalAt, Plane.cpp:24-26
void ref_plane_hit(int n, const float* p, const float* normal, const float* origin, const float* dir, int* ok, float* t0, float* normalOut) {
    for (int i = 0; i < n; i++) {
        Plane pl(v3(p + 3 * i), v3(normal + 3 * i), kNoMaterial);
        float t1 = 0;
        ok[i] = pl.hit(v3(origin + 3 * i), v3(dir + 3 * i), t0[i], t1) ? 1 : 0;
        if (normalOut) put(normalOut + 3 * i, pl.normalAt(vec3()));
    }
}

// Triangle::hit, Triangle.cpp:20-42 (the stand-alone triangle object; no scene of the reference holds one)
void ref_triangle_hit(int n, const float* a, const float* b, const float* c, const float* origin, const float* dir, int* ok, float* t0) {
    for (int i = 0; i < n; i++) {
        Triangle tri(v3(a + 3 * i), v3(b + 3 * i), v3(c + 3 * i), kNoMaterial);
        float t1 = 0;
        ok[i] = tri.hit(v3(origin + 3 * i), v3(dir + 3 * i), t0[i], t1) ? 1 : 0;
    }
}

// vec3.h:26, 102-117: length(a), dot(a, b), cross(a, b), normalise(a), reflect(a, b); and utils.h:102-104 radians(a.x)
void ref_vec3_ops(int n, const float* a, const float* b, float* length, float* dotOut, float* crossOut, float* normalised, float* reflected, float* rad) {
    for (int i = 0; i < n; i++) {
        const vec3 x = v3(a + 3 * i), y = v3(b + 3 * i);
        length[i] = x.length();
        dotOut[i] = dot(x, y);
        put(crossOut + 3 * i, cross(x, y));
        put(normalised + 3 * i, normalise(x));
        put(reflected + 3 * i, reflect(x, y));
        rad[i] = radians(a[3 * i]);
    }
}

// Camera::Camera, Camera.cpp:5-20: the 84 bytes of the object the reference hands to its kernel
void ref_camera_make(const float* position, float fov, float aspectRatio, float yaw, float pitch, unsigned char* out84) {
    Camera cam(v3(position), fov, aspectRatio, yaw, pitch);
    std::memcpy(out84, (const void*)&cam, sizeof(Camera));
}

// Camera::move, Camera.cpp:65-83 (moves: Camera::CameraMove values) and Camera::mouseMovement, Camera.cpp:40-52 (mouse: pairs
// of offsets applied after the moves)
void ref_camera_drive(unsigned char* cam84, const int* moves, int nMoves, const double* mouse, int nMouse) {
    Camera cam = cameraFromBytes(cam84);
    for (int i = 0; i < nMoves; i++) cam.move((Camera::CameraMove)moves[i]);
    for (int i = 0; i < nMouse; i++) cam.mouseMovement(mouse[2 * i], mouse[2 * i + 1]);
    std::memcpy(cam84, (const void*)&cam, sizeof(Camera));
}

// Per pixel of the window [x0, x1) x [y0, y1): raw = Camera::rasterToCameraSpace(float(x + offX), float(y + offY), width, height)
// (Camera.cpp:22-38, called as in kernels.cu:337,355) and dir = normalise(raw) (vec3.h:111-113, kernels.cu:338,356).
// Outputs are (y1 - y0) x (x1 - x0) x 3 floats; either may be null.
void ref_camera_rays(const unsigned char* cam84, int width, int height, int x0, int y0, int x1, int y1, double offX, double offY, float* raw, float* dir) {
    Camera cam = cameraFromBytes(cam84);
    size_t k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, k++) {
            vec3 p = cam.rasterToCameraSpace(float(x + offX), float(y + offY), width, height);
            if (raw) put(raw + 3 * k, p);
            if (dir) put(dir + 3 * k, normalise(p));
        }
}

// For every primary ray of a frame (pixel centres, as kernels.cu:355-356) and every sphere: Sphere::hit's verdict and t0.
// ok: height x width x count bytes, t0: height x width x count floats (left untouched where ok = 0).
void ref_frame_spheres(const unsigned char* cam84, int width, int height, int count, const float* centre, const float* radius, unsigned char* ok, float* t0) {
    Camera cam = cameraFromBytes(cam84);
    std::vector<Sphere> spheres;
    for (int i = 0; i < count; i++) spheres.emplace_back(v3(centre + 3 * i), radius[i], kNoMaterial);
    size_t k = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            vec3 d = normalise(cam.rasterToCameraSpace(float(x + 0.5), float(y + 0.5), width, height));
            for (int i = 0; i < count; i++, k++) {
                float t1 = 0;
                ok[k] = spheres[(size_t)i].hit(cam.getPosition(), d, t0[k], t1) ? 1 : 0;
            }
        }
}

// The same for planes (Plane::hit).
void ref_frame_planes(const unsigned char* cam84, int width, int height, int count, const float* p, const float* normal, unsigned char* ok, float* t0) {
    Camera cam = cameraFromBytes(cam84);
    std::vector<Plane> planes;
    for (int i = 0; i < count; i++) planes.emplace_back(v3(p + 3 * i), v3(normal + 3 * i), kNoMaterial);
    size_t k = 0;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            vec3 d = normalise(cam.rasterToCameraSpace(float(x + 0.5), float(y + 0.5), width, height));
            for (int i = 0; i < count; i++, k++) {
                float t1 = 0;
                ok[k] = planes[(size_t)i].hit(cam.getPosition(), d, t0[k], t1) ? 1 : 0;
            }
        }
}

// ---- objl::Loader (OBJ_Loader.cpp:183-465); same shapes as the host mirror's rth_loader_* so one comparison serves both ----

void* ref_loader_load(const char* path) {
    QuietCout quiet;
    LoaderHolder* h = new LoaderHolder();
    h->ok = h->loader.LoadFile(path);
    return h;
}
void ref_loader_free(void* h) { delete static_cast<LoaderHolder*>(h); }
// out5: LoadFile's return value, LoadedVertices, LoadedIndices, LoadedMeshes, LoadedMaterials
void ref_loader_counts(void* h, int* out5) {
    LoaderHolder* l = static_cast<LoaderHolder*>(h);
    out5[0] = l->ok ? 1 : 0; out5[1] = (int)l->loader.LoadedVertices.size(); out5[2] = (int)l->loader.LoadedIndices.size();
    out5[3] = (int)l->loader.LoadedMeshes.size(); out5[4] = (int)l->loader.LoadedMaterials.size();
}
void ref_loader_arrays(void* h, float* vertices8, unsigned int* indices) {
    LoaderHolder* l = static_cast<LoaderHolder*>(h);
    dumpVertices(l->loader.LoadedVertices, vertices8);
    for (size_t i = 0; i < l->loader.LoadedIndices.size(); i++) indices[i] = l->loader.LoadedIndices[i];
}
void ref_loader_mesh(void* h, int m, int* counts2, char* name, char* materialName, int cap, float* material13, float* vertices8) {
    const objl::Mesh& mesh = static_cast<LoaderHolder*>(h)->loader.LoadedMeshes[(size_t)m];
    counts2[0] = (int)mesh.Vertices.size(); counts2[1] = (int)mesh.Indices.size();
    copyName(name, cap, mesh.MeshName);
    copyName(materialName, cap, mesh.MeshMaterial.name);
    dumpMaterial(mesh.MeshMaterial, material13);
    if (vertices8) dumpVertices(mesh.Vertices, vertices8);
}
void ref_loader_material(void* h, int k, char* name, char* mapKd, int cap, float* material13) {
    const objl::Material& mt = static_cast<LoaderHolder*>(h)->loader.LoadedMaterials[(size_t)k];
    copyName(name, cap, mt.name);
    copyName(mapKd, cap, mt.map_Kd);
    dumpMaterial(mt, material13);
}

}  // extern "C"
