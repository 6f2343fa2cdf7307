// Host mirror of Triangle (Triangle.h:6-20, Triangle.cpp:9-17).  The reference's Scene has no triangle array
// (RayTracer.h:17-34), so a stand-alone Triangle can be constructed but is never rendered; meshes go through Model.
#pragma once
#include "Object.h"

class Triangle : public Object {
public:
    vec3 a, b, c;

    Triangle() {}
    Triangle(vec3 a, vec3 b, vec3 c, Material mat, bool debug = false, ObjectType objectType = Diffuse) : a(a), b(b), c(c) {
        this->mat = mat; this->debug = debug; this->objectType = objectType; objectName = Triangle_t;
    }
};
