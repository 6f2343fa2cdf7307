// Host mirror of Sphere (Sphere.h:6-21, Sphere.cpp:4-15).
#pragma once
#include "Object.h"

class Sphere : public Object {
public:
    float radius = 0.0f;

    Sphere() {}
    Sphere(vec3 center, float radius, Material mat, bool debug = false, ObjectType objectType = Diffuse) : radius(radius) {
        position = center; this->mat = mat; this->debug = debug; this->objectType = objectType; objectName = Sphere_t;
    }
};
