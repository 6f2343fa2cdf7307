// Host mirror of the reference's shared enums and structs (utils.h:14-67, 86-104).
#pragma once
#include <string>

#include "vec3.h"

enum ObjectName { Sphere_t, Plane_t, Box_t, Triangle_t, Model_t, AABB_t };
enum ObjectType { Diffuse, Reflect };
enum RayType { PrimaryRay, ShadowRay, ReflectRay };

struct Light {
    vec3 position;
    vec3 colour;
    vec3 uvec;
    int usteps;
    vec3 vvec;
    int vsteps;
    int samples;
};

struct Material {
    vec3 ambientColour;
    vec3 diffuseColour;
    float ambient;
    float diffuse;
    float specular;
    float shininess;
};

// Same 76-byte layout as rt_vertex, so a Vertex array can be handed to the C ABI unchanged.
struct Vertex {
    vec3 position;
    vec3 textureCoords;
    vec3 normal;
    int materialIndex;
    vec3 ambient;
    vec3 diffuse;
    vec3 specular;
};

struct Hit {
    float t = 999;
    Material mat = {};
    vec3 hitPoint;
    vec3 normal;
    vec3 objectPos;
    ObjectType objectType = Diffuse;
    ObjectName objectName = Sphere_t;
    bool debug = false;
};

template <typename T> T min(T a, T b) { return (a < b) ? a : b; }
template <typename T> T max(T a, T b) { return (a > b) ? a : b; }
template <typename T> void swap(T& a, T& b) { T t = a; a = b; b = t; }

inline float radians(float angle) { return (float)(angle * 0.01745329251994329576923690768489); }
