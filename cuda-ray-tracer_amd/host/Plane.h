// Host mirror of Plane (Plane.h:6-22, Plane.cpp:3-14).  As in the reference, the constructor ignores its debug
// argument: a plane is never a debug object (Plane.cpp:11).
#pragma once
#include "Object.h"

class Plane : public Object {
public:
    vec3 n;

    Plane() {}
    Plane(vec3 p, vec3 n, Material mat, bool /*debug*/ = false, ObjectType objectType = Diffuse) : n(n) {
        position = p; this->mat = mat; this->debug = false; this->objectType = objectType; objectName = Plane_t;
    }
};
