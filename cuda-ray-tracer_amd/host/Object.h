// Host mirror of Object (Object.h:8-25): the fields every renderable carries.  The reference declares hit() and
// normalAt() here for its device code; on this side of the C ABI intersection runs only in the gfx950 kernels.
#pragma once
#include "utils.h"

class Object {
public:
    vec3 position;
    Material mat = {};
    ObjectType objectType = Diffuse;
    ObjectName objectName = Sphere_t;
    bool debug = false;
};
