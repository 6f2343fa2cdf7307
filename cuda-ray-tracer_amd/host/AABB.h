// Host mirror of AABB (AABB.h:6-26, AABB.cpp:3-34, 97-101): a renderable box and the bounding volume of models.
#pragma once
#include <vector>

#include "Object.h"

class AABB : public Object {
public:
    vec3 min, max;
    int indicesCount = 0;
    std::vector<int> includedModelIndices;   // the reference's managed int array (AABB.h:24-25)

    AABB() { objectName = AABB_t; }   // min = max = (0,0,0): AABB.cpp:3-6
    AABB(vec3 pos, float size, Material mat, bool debug = false, ObjectType objectType = Diffuse)
        : min(pos - (size / 2)), max(pos + (size / 2)) {
        this->mat = mat; this->debug = debug; this->objectType = objectType; objectName = AABB_t;
    }
    AABB(vec3 min, vec3 max, Material mat, bool debug = false, ObjectType objectType = Diffuse) : min(min), max(max) {
        this->mat = mat; this->debug = debug; this->objectType = objectType; objectName = AABB_t;
    }

    // AABB.cpp:26-34: a point outside pushes the face 0.1 beyond it (double arithmetic, then narrowed)
    void extendBy(vec3 point) {
        for (int a = 0; a < 3; a++) {
            if (point[a] < min[a]) min[a] = (float)(point[a] - 0.1);
            if (point[a] > max[a]) max[a] = (float)(point[a] + 0.1);
        }
    }
    void setupIndices(int count) { indicesCount = count; includedModelIndices.assign((size_t)count, 0); }
};
