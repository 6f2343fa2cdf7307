/*
 * oracle.c — CPU restatement of the reference ray tracer's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under the product package may include,
 * link, import or execute this file; it exists so that tests, the smoke check
 * and the bench's cpu_baseline leg have an independent statement of what the
 * reference computes.
 *
 * PARITY: PARTLY PINNED.  The reference ships no tests, golden images or
 * fixtures for this path.  What compiles of it with g++ is built from the
 * checkout as oracle/_ref/ (recipe: oracle/build_ref.sh; the CUDA headers come
 * from the image's triton package, glew and glm from the reference's own
 * includes/) and this file is checked against it bit for bit:
 *   pinned   solve_quadratic, sphere_hit, plane_hit, triangle_object_hit, the
 *            vec3 helpers, the camera basis and raster_to_camera_space (against
 *            utils.cpp, Sphere.cpp, Plane.cpp, Triangle.cpp, vec3.h, Camera.cpp),
 *            triangle_intersect with model_hit's no-box branch and the attribute
 *            interpolation (against Model.cpp:135-159, 318-356), in
 *            tests/test_ref_pins.py and the reference-generated fixtures
 *            tests/golden/refgen_*.npz;
 *   unpinned everything that restates kernels.cu (it needs <curand_kernel.h>,
 *            which the image lacks, and binds temporaries to `vec3&` at
 *            kernels.cu:67,130,152,188,224, which only MSVC accepts) and
 *            AABB.cpp (the same extension at AABB.cpp:90): wang_hash, lighting,
 *            the trace loops, ray_cast and the reflection casts, render_pixel,
 *            aabb_hit / aabb_normal_at / aabb_extend_by, model_hit's box-gated
 *            branches, the model bake (its arithmetic sits in Model::Model,
 *            which constructs AABB members).  Those follow the reference's
 *            source text line by line in meaning, each function citing the lines
 *            it restates, and have not been checked against its outputs.
 *
 * Arithmetic contract: IEEE-754 binary32 for every operation the reference
 * performs in float, evaluated in the reference's order, no contraction
 * (build with -ffp-contract=off, no -ffast-math), double only where the
 * reference's literals promote to double.  Float->int and float->u8 casts use
 * the saturating, NaN->0 semantics of the GPU the reference targets.
 *
 * Plain C (C11), no dependencies beyond libm and (optionally) OpenMP.
 */
#include "rt_types.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef rt_vec3 v3;

#define FAR_T 999.0f   /* utils.h:49, kernels.cu:90,122, Model.cpp:277,299 */

/* ------------------------------------------------------------------ vec3.h */

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
/* vec3.h:66-68 */
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
/* vec3.h:74-76 */
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
/* vec3.h:78-80 */
static inline v3 v3_subs(v3 a, float n) { return v3_make(a.x - n, a.y - n, a.z - n); }
/* vec3.h:70-72 */
static inline v3 v3_adds(v3 a, float n) { return v3_make(a.x + n, a.y + n, a.z + n); }
/* vec3.h:82-84 */
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
/* vec3.h:86-92 (both operand orders compute v[i] * n) */
static inline v3 v3_scale(v3 a, float n) { return v3_make(a.x * n, a.y * n, a.z * n); }
/* vec3.h:98-100 */
static inline v3 v3_divs(v3 a, float n) { return v3_make(a.x / n, a.y / n, a.z / n); }
/* vec3.h:20 */
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* vec3.h:102-104 — sum is (x + y) + z */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec3.h:106-110 */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.h:24 */
static inline float v3_length(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* vec3.h:112-114 — three divisions by the length */
static inline v3 v3_normalise(v3 a) { return v3_divs(a, v3_length(a)); }
/* vec3.h:116-118 — a - ((b * dot) * (2,2,2)) */
static inline v3 v3_reflect(v3 a, v3 b) {
    return v3_sub(a, v3_mul(v3_scale(b, v3_dot(a, b)), v3_make(2.0f, 2.0f, 2.0f)));
}
/* vec3.h:41-43 */
static inline int v3_ne(v3 a, v3 b) { return !((a.x == b.x) && (a.y == b.y) && (a.z == b.z)); }

/* utils.h:86-93 — the comparison-based min/max (NaN falls to the second operand). */
static inline float f_min(float x, float y) { return (x < y) ? x : y; }
static inline float f_max(float x, float y) { return (x > y) ? x : y; }

/* float -> int, round toward zero, saturating, NaN -> 0 (device cast semantics). */
static inline int f2i_rz(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

/* ------------------------------------------------------------- utils.cpp */

/* utils.cpp:10-32 */
static int solve_quadratic(float a, float b, float c, float* x0, float* x1) {
    float discriminant = b * b - 4 * a * c;
    if (discriminant < 0) {
        return 0;
    } else if (discriminant == 0) {
        /* -0.5 is a double literal: the whole expression is evaluated in double */
        *x0 = *x1 = (float)(-0.5 * (double)b / (double)a);
    } else {
        float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(discriminant)))
                          : (float)(-0.5 * (double)(b - sqrtf(discriminant)));
        *x0 = q / a;
        *x1 = c / q;
    }
    if (*x0 > *x1) { float tmp = *x0; *x0 = *x1; *x1 = tmp; }
    return 1;
}

/* kernels.cu:11-18 */
static float wang_hash(unsigned int seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return (float)seed / 4294967296.0f;
}

/* ------------------------------------------------------------ primitives */

/* Sphere.cpp:17-44 */
static int sphere_hit(const rt_sphere* s, v3 o, v3 d, float* t0, float* t1) {
    float a0, a1;
    v3 L = v3_sub(o, s->position);
    float a = v3_dot(d, d);
    float b = 2 * v3_dot(d, L);
    float c = v3_dot(L, L) - s->radius * s->radius;
    if (!solve_quadratic(a, b, c, &a0, &a1)) return 0;
    if (a0 < 0) { a0 = a1; if (a0 < 0) return 0; }
    if (a1 < 0) { a1 = a0; if (a1 < 0) return 0; }
    if (a0 > a1) { float tmp = a1; a1 = a0; a0 = tmp; }
    *t0 = a0; *t1 = a1;
    return 1;
}

/* Plane.cpp:16-22 */
static int plane_hit(const rt_plane* p, v3 o, v3 d, float* t0) {
    float denominator = v3_dot(p->n, d);
    v3 v = v3_sub(p->position, o);
    *t0 = v3_dot(v, p->n) / denominator;
    return (*t0 >= 0);
}

/* AABB.cpp:36-84 */
static int aabb_hit(v3 bmin, v3 bmax, v3 o, v3 d, float* t0, float* t1) {
    float tmp;
    float tmin = (bmin.x - o.x) / d.x;
    float tmax = (bmax.x - o.x) / d.x;
    if (tmin > tmax) { tmp = tmin; tmin = tmax; tmax = tmp; }

    float tmin_y = (bmin.y - o.y) / d.y;
    float tmax_y = (bmax.y - o.y) / d.y;
    if (tmin_y > tmax_y) { tmp = tmin_y; tmin_y = tmax_y; tmax_y = tmp; }

    if ((tmin > tmax_y) || (tmin_y > tmax)) return 0;
    if (tmin_y > tmin) tmin = tmin_y;
    if (tmax_y < tmax) tmax = tmax_y;

    float tmin_z = (bmin.z - o.z) / d.z;
    float tmax_z = (bmax.z - o.z) / d.z;
    if (tmin_z > tmax_z) { tmp = tmin_z; tmin_z = tmax_z; tmax_z = tmp; }

    if ((tmin > tmax_z) || (tmin_z > tmax)) return 0;
    if (tmin_z > tmin) tmin = tmin_z;
    if (tmax_z < tmax) tmax = tmax_z;

    if (tmin < 0) { tmin = tmax; if (tmin < 0) return 0; }
    if (tmax < 0) { tmax = tmin; if (tmax < 0) return 0; }
    if (tmin > tmax) { tmp = tmin; tmin = tmax; tmax = tmp; }

    *t0 = tmin; *t1 = tmax;
    return 1;
}

/* AABB.cpp:86-95 (with utils.cpp:48-50 for abs) */
static v3 aabb_normal_at(v3 bmin, v3 bmax, v3 point) {
    v3 centre = v3_divs(v3_add(bmin, bmax), 2.0f);
    v3 p = v3_sub(point, centre);
    v3 e = v3_sub(bmin, bmax);
    v3 d = v3_divs(v3_make(fabsf(e.x), fabsf(e.y), fabsf(e.z)), 2.0f);
    float bias = 1.00001f;
    v3 n = v3_make((float)f2i_rz(p.x / d.x * bias), (float)f2i_rz(p.y / d.y * bias), (float)f2i_rz(p.z / d.z * bias));
    return v3_normalise(n);
}

/* Model.cpp:135-159 */
static int triangle_intersect(v3 p0, v3 p1, v3 p2, v3 o, v3 d, float* t, float* u, float* v, int cull) {
    v3 e1 = v3_sub(p1, p0);
    v3 e2 = v3_sub(p2, p0);
    v3 pvec = v3_cross(d, e2);
    float det = v3_dot(e1, pvec);

    if (cull && (double)det >= -0.01) return 0;   /* -0.01 is a double literal */
    if (fabsf(det) <= 0) return 0;

    float invDet = 1 / det;

    v3 tvec = v3_sub(o, p0);
    *u = v3_dot(tvec, pvec) * invDet;
    if (*u < 0 || *u > 1) return 0;

    v3 qvec = v3_cross(tvec, e1);
    *v = v3_dot(d, qvec) * invDet;
    if (*v < 0 || *u + *v > 1) return 0;

    *t = v3_dot(e2, qvec) * invDet;
    return 1;
}

/* Triangle.cpp:20-42 — the standalone Triangle object (never placed in a scene). */
static int triangle_object_hit(v3 a, v3 b, v3 c, v3 o, v3 d, float* t0) {
    v3 e1 = v3_sub(b, a);
    v3 e2 = v3_sub(c, a);
    v3 pvec = v3_cross(d, e2);
    float det = v3_dot(e1, pvec);
    if ((double)fabsf(det) < 0.001) return 0;
    float invDet = 1 / det;
    v3 tvec = v3_sub(o, a);
    float u = v3_dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return 0;
    v3 qvec = v3_cross(tvec, e1);
    float v = v3_dot(d, qvec) * invDet;
    if (v < 0 || u + v > 1) return 0;
    *t0 = v3_dot(e2, qvec) * invDet;
    return 1;
}

/* ----------------------------------------------------------- Model::hit */

typedef struct {
    v3 normal, ambient, diffuse;   /* the interpolated members the shader reads */
    int tri;                       /* index of the winning triangle (i / 3 of Model.cpp:282) */
} hit_vertex;

/* Model.cpp:300-310 — weights (u, v, 1-u-v) applied to (v0, v1, v2) in that order. */
static v3 interp3(float u, float v, float w, v3 a, v3 b, v3 c) {
    return v3_add(v3_add(v3_scale(a, u), v3_scale(b, v)), v3_scale(c, w));
}

static void fill_hit_vertex(const rt_model* m, int i0, int i1, int i2, float u, float v, int tri, hit_vertex* hv) {
    const rt_vertex* a = &m->vertices[i0];
    const rt_vertex* b = &m->vertices[i1];
    const rt_vertex* c = &m->vertices[i2];
    float w = 1 - u - v;
    hv->normal = interp3(u, v, w, a->normal, b->normal, c->normal);
    hv->ambient = interp3(u, v, w, a->ambient, b->ambient, c->ambient);
    hv->diffuse = interp3(u, v, w, a->diffuse, b->diffuse, c->diffuse);
    hv->tri = tri;
}

/*
 * Model.cpp:161-357.  accelerate / accelerateTwice / useOctree select the mode
 * in the reference's priority order: octree, eight boxes, whole-model box, none.
 */
static int model_hit(const rt_model* m, v3 o, v3 d, float* t0, hit_vertex* hv,
                     int accelerate, int accelerateTwice, int useOctree, int cull, uint64_t* tri_tests) {
    if (useOctree) {
        return 0;                                           /* Model.cpp:163-219 */
    } else if (accelerateTwice) {                           /* Model.cpp:220-274 */
        hit_vertex best; float bestT = FAR_T;
        memset(&best, 0, sizeof best);
        for (int b = 0; b < 8; b++) {
            const rt_chunk_box* bb = &m->boundingBoxes[b];
            float bt0, bt1;
            if (aabb_hit(bb->min, bb->max, o, d, &bt0, &bt1)) {
                float tClosest = FAR_T, u = 0, v = 0; int w0 = 0, w1 = 0, w2 = 0, wj = 0;
                for (int j = 0; j < bb->indicesCount - 3; j += 3) {
                    float tt, uu, vv;
                    int i0 = bb->includedModelIndices[j], i1 = bb->includedModelIndices[j + 1], i2 = bb->includedModelIndices[j + 2];
                    if (tri_tests) (*tri_tests)++;
                    if (triangle_intersect(m->vertices[i0].position, m->vertices[i1].position, m->vertices[i2].position, o, d, &tt, &uu, &vv, cull)) {
                        if (tt < tClosest && tt > 0) { tClosest = tt; w0 = i0; w1 = i1; w2 = i2; u = uu; v = vv; wj = j; }
                    }
                }
                if (tClosest != FAR_T) {
                    if (tClosest < bestT) {
                        bestT = tClosest;
                        fill_hit_vertex(m, w0, w1, w2, u, v, b * (bb->indicesCount / 3) + wj / 3, &best);
                    }
                }
            }
        }
        if (bestT != FAR_T) { *hv = best; *t0 = bestT; return 1; }
        return 0;
    } else if (accelerate) {                                /* Model.cpp:275-317 */
        float t1;
        if (aabb_hit(m->bbMin, m->bbMax, o, d, t0, &t1)) {
            float tClosest = FAR_T, u = 0, v = 0; int w0 = 0, w1 = 0, w2 = 0, wi = 0;
            for (int i = 0; i < m->indicesCount - 3; i += 3) {
                float tt, uu, vv;
                int i0 = m->indices[i], i1 = m->indices[i + 1], i2 = m->indices[i + 2];
                if (tri_tests) (*tri_tests)++;
                if (triangle_intersect(m->vertices[i0].position, m->vertices[i1].position, m->vertices[i2].position, o, d, &tt, &uu, &vv, cull)) {
                    if (tt < tClosest && tt > 0) { tClosest = tt; w0 = i0; w1 = i1; w2 = i2; u = uu; v = vv; wi = i; }
                }
            }
            if (tClosest != FAR_T) {
                fill_hit_vertex(m, w0, w1, w2, u, v, wi / 3, hv);
                *t0 = tClosest;
                return 1;
            }
        }
        return 0;
    } else {                                                /* Model.cpp:318-356: no box, no t > 0 test */
        float tClosest = FAR_T, u = 0, v = 0; int w0 = 0, w1 = 0, w2 = 0, wi = 0;
        for (int i = 0; i < m->indicesCount - 3; i += 3) {
            float tt, uu, vv;
            int i0 = m->indices[i], i1 = m->indices[i + 1], i2 = m->indices[i + 2];
            if (tri_tests) (*tri_tests)++;
            if (triangle_intersect(m->vertices[i0].position, m->vertices[i1].position, m->vertices[i2].position, o, d, &tt, &uu, &vv, cull)) {
                if (tt < tClosest) { tClosest = tt; w0 = i0; w1 = i1; w2 = i2; u = uu; v = vv; wi = i; }
            }
        }
        if (tClosest != FAR_T) {
            fill_hit_vertex(m, w0, w1, w2, u, v, wi / 3, hv);
            *t0 = tClosest;
            return 1;
        }
        return 0;
    }
}

/* ------------------------------------------------------------- tracing */

/* utils.h:48-57 (Hit) plus the identity of the winning object, which the reference does not keep. */
typedef struct {
    float t;
    rt_material mat;
    v3 hitPoint, normal;
    int objectType, objectName, debug;
    int index, tri;
} o_hit;

static void hit_reset(o_hit* h) {
    memset(h, 0, sizeof *h);
    h->t = FAR_T; h->objectType = RT_DIFFUSE; h->objectName = -1; h->index = -1; h->tri = -1;
}

typedef struct {
    const rt_scene_desc* scene;
    const rt_camera* cam;
    const rt_config* cfg;
    uint64_t rays, tri_tests;
} o_ctx;

static const v3 IGNORE_POS = { -999.0f, -999.0f, -999.0f };   /* kernels.cu:67 default argument */

/* kernels.cu:67-96 for T = Sphere */
static int trace_spheres(const o_ctx* c, v3 o, v3 d, int rayType, o_hit* hit) {
    const rt_scene_desc* s = c->scene;
    for (int i = 0; i < s->sphereCount; i++) {
        const rt_sphere* ob = &s->spheres[i];
        if (v3_ne(ob->position, IGNORE_POS)) {
            float t0, t1;
            if (sphere_hit(ob, o, d, &t0, &t1)) {
                if (rayType == RT_SHADOW_RAY) { if (!ob->debug) return 1; continue; }
                if (rayType == RT_REFLECT_RAY && ob->debug) continue;
                if (t0 < hit->t) {
                    v3 hp = v3_add(o, v3_scale(d, t0));
                    hit->t = t0; hit->mat = ob->mat; hit->hitPoint = hp;
                    hit->normal = v3_sub(hp, ob->position);            /* Sphere.cpp:46-48 */
                    hit->objectType = ob->objectType; hit->objectName = RT_SPHERE_T; hit->debug = ob->debug;
                    hit->index = i; hit->tri = -1;
                }
            }
        }
    }
    if (hit->t != FAR_T) { hit->normal = v3_normalise(hit->normal); return 1; }
    return 0;
}

/* kernels.cu:67-96 for T = Plane */
static int trace_planes(const o_ctx* c, v3 o, v3 d, int rayType, o_hit* hit) {
    const rt_scene_desc* s = c->scene;
    for (int i = 0; i < s->planeCount; i++) {
        const rt_plane* ob = &s->planes[i];
        if (v3_ne(ob->position, IGNORE_POS)) {
            float t0;
            if (plane_hit(ob, o, d, &t0)) {
                if (rayType == RT_SHADOW_RAY) { if (!ob->debug) return 1; continue; }
                if (rayType == RT_REFLECT_RAY && ob->debug) continue;
                if (t0 < hit->t) {
                    v3 hp = v3_add(o, v3_scale(d, t0));
                    hit->t = t0; hit->mat = ob->mat; hit->hitPoint = hp;
                    hit->normal = v3_neg(ob->n);                      /* Plane.cpp:24-26 */
                    hit->objectType = ob->objectType; hit->objectName = RT_PLANE_T; hit->debug = ob->debug;
                    hit->index = i; hit->tri = -1;
                }
            }
        }
    }
    if (hit->t != FAR_T) { hit->normal = v3_normalise(hit->normal); return 1; }
    return 0;
}

/* kernels.cu:67-96 for T = AABB (Object::position of a box is never assigned, so the ignore test always passes) */
static int trace_boxes(const o_ctx* c, v3 o, v3 d, int rayType, o_hit* hit) {
    const rt_scene_desc* s = c->scene;
    for (int i = 0; i < s->AABBCount; i++) {
        const rt_aabb* ob = &s->AABBs[i];
        float t0, t1;
        if (aabb_hit(ob->min, ob->max, o, d, &t0, &t1)) {
            if (rayType == RT_SHADOW_RAY) { if (!ob->debug) return 1; continue; }
            if (rayType == RT_REFLECT_RAY && ob->debug) continue;
            if (t0 < hit->t) {
                v3 hp = v3_add(o, v3_scale(d, t0));
                hit->t = t0; hit->mat = ob->mat; hit->hitPoint = hp;
                hit->normal = aabb_normal_at(ob->min, ob->max, hp);
                hit->objectType = ob->objectType; hit->objectName = RT_AABB_T; hit->debug = ob->debug;
                hit->index = i; hit->tri = -1;
            }
        }
    }
    if (hit->t != FAR_T) { hit->normal = v3_normalise(hit->normal); return 1; }
    return 0;
}

/* kernels.cu:98-128 */
static int trace_models(o_ctx* c, v3 o, v3 d, int rayType, o_hit* hit) {
    const rt_scene_desc* s = c->scene;
    const rt_config* g = c->cfg;
    for (int i = 0; i < s->modelCount; i++) {
        const rt_model* ob = &s->models[i];
        float t0; hit_vertex hv;
        if (model_hit(ob, o, d, &t0, &hv, g->boundingBox, g->eightBoundingBoxes, g->octree, g->cullBackTriangles, &c->tri_tests)) {
            if (rayType == RT_SHADOW_RAY) { if (!ob->debug) return 1; continue; }
            if (rayType == RT_REFLECT_RAY && ob->debug) continue;
            if (t0 < hit->t) {
                v3 hp = v3_add(o, v3_scale(d, t0));
                hit->t = t0; hit->mat = ob->mat;
                hit->mat.ambientColour = hv.ambient;                  /* kernels.cu:115 */
                hit->mat.diffuseColour = hv.diffuse;                  /* kernels.cu:116 */
                hit->hitPoint = hp; hit->normal = hv.normal;
                hit->objectType = ob->objectType; hit->objectName = RT_MODEL_T; hit->debug = ob->debug;
                hit->index = i; hit->tri = hv.tri;
            }
        }
    }
    if (hit->t != FAR_T) { hit->normal = v3_normalise(hit->normal); return 1; }
    return 0;
}

/* -------------------------------------------------------------- shading */

/* kernels.cu:44-46 / 295-297 — a point on the light's sphere from two hashed numbers */
/* The reference's sin / cos / acos on floats are evaluated in double and rounded to float once: the correctly rounded
 * float result, which is what independent math libraries (this one, the GPU's, the reference's) can agree on. */
static int g_libm_float_trig = 0;
static float cr_sinf(float x) { return g_libm_float_trig ? sinf(x) : (float)sin((double)x); }
static float cr_cosf(float x) { return g_libm_float_trig ? cosf(x) : (float)cos((double)x); }
static float cr_acosf(float x) { return g_libm_float_trig ? acosf(x) : (float)acos((double)x); }
/* Test switch: evaluate the three functions with this machine's single-precision libm routines, i.e. the reference's own
 * formulation (sin / cos / acos on float arguments, kernels.cu:44-46, 295-297) as one particular library computes it.  Used to
 * bound how far the correctly rounded definition can move a frame away from such a build.  Not thread-safe; set between renders. */
void oracle_set_libm_float_trig(int on) { g_libm_float_trig = on != 0; }

static v3 light_sample_offset(unsigned int seed, const rt_config* g) {
    float radius = (float)g->softShadowRadius;
    float theta = 2 * 3.141592654f * wang_hash(seed);
    float phi = cr_acosf(2 * wang_hash(seed * 16807u) - 1);
    return v3_make(radius * cr_sinf(phi) * cr_cosf(theta), radius * cr_sinf(phi) * cr_sinf(theta), radius * cr_cosf(phi));
}

/* kernels.cu:42, 284 — (x * 1000 + y) * 1973 + i * 9277 in wrapping 32-bit arithmetic */
static unsigned int sample_seed(int x, int y, int i) {
    return ((unsigned int)x * 1000u + (unsigned int)y) * 1973u + (unsigned int)i * 9277u;
}

/* kernels.cu:20-65 */
static v3 lighting(rt_material mat, v3 lightPos, v3 lightIntensity, v3 point, v3 eye, v3 normal,
                   const rt_config* g, int px, int py) {
    v3 colour = v3_make(0, 0, 0);
    v3 L = v3_normalise(v3_sub(lightPos, point));

    if (g->ambientLighting) {
        colour = v3_scale(v3_mul(lightIntensity, mat.ambientColour), mat.ambient);
    }
    if (g->diffuseLighting) {
        float NdotL = f_max(v3_dot(normal, L), 0.0f);
        colour = v3_add(colour, (NdotL > 0.0)
                                    ? v3_scale(v3_scale(v3_mul(mat.diffuseColour, lightIntensity), mat.diffuse), NdotL)
                                    : v3_make(0, 0, 0));
    }
    if (g->specularLighting) {
        if (g->areaLightSpecularEffect) {
            for (int i = 0; i < g->softShadowNum; i++) {
                v3 ps = light_sample_offset(sample_seed(px, py, i), g);
                v3 Ls = v3_normalise(v3_sub(v3_add(lightPos, ps), point));
                v3 v = v3_normalise(eye);
                v3 r = v3_normalise(v3_neg(v3_reflect(Ls, normal)));
                float RdotV = f_max(v3_dot(r, v), 0.0f);
                colour = v3_add(colour, v3_scale(v3_scale(lightIntensity, mat.specular), powf(RdotV, mat.shininess)));
            }
        } else {
            v3 v = v3_normalise(eye);
            v3 r = v3_normalise(v3_neg(v3_reflect(L, normal)));
            float RdotV = f_max(v3_dot(r, v), 0.0f);
            colour = v3_add(colour, v3_scale(v3_scale(lightIntensity, mat.specular), powf(RdotV, mat.shininess)));
        }
    }
    return colour;
}

/* One (origin, direction) sphere-only shadow test: kernels.cu:140-144, 162-166, 212-216 */
static int sphere_shadow(o_ctx* c, const o_hit* h) {
    const rt_config* g = c->cfg;
    o_hit sh; hit_reset(&sh);
    c->rays++;
    return trace_spheres(c, v3_add(h->hitPoint, v3_scale(h->normal, g->shadowBias)),
                         v3_normalise(v3_sub(c->scene->lights[0].position, h->hitPoint)), RT_SHADOW_RAY, &sh);
}

/* kernels.cu:130-150 */
static v3 reflection_cast3(o_ctx* c, v3 o, v3 d, int px, int py) {
    const rt_scene_desc* s = c->scene; const rt_config* g = c->cfg;
    o_hit h; hit_reset(&h);
    c->rays++;
    int a = trace_spheres(c, o, d, RT_REFLECT_RAY, &h);
    int b = g->reflectPlanes ? trace_planes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int e = g->renderAABBs ? trace_boxes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int f = g->renderModels ? trace_models(c, o, d, RT_REFLECT_RAY, &h) : 0;
    if (a || b || e || f) {
        v3 col = lighting(h.mat, s->lights[0].position, s->lights[0].colour, h.hitPoint, c->cam->position, v3_normalise(h.normal), g, px, py);
        if (g->renderHardShadows) {
            if (sphere_shadow(c, &h)) col = v3_scale(col, g->shadowIntensity);
        }
        return col;
    }
    return g->backgroundCol;
}

/* kernels.cu:152-186 */
static v3 reflection_cast2(o_ctx* c, v3 o, v3 d, int px, int py) {
    const rt_scene_desc* s = c->scene; const rt_config* g = c->cfg;
    int depth = 1;
    o_hit h; hit_reset(&h);
    c->rays++;
    int a = trace_spheres(c, o, d, RT_REFLECT_RAY, &h);
    int b = g->reflectPlanes ? trace_planes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int e = g->renderAABBs ? trace_boxes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int f = g->renderModels ? trace_models(c, o, d, RT_PRIMARY_RAY, &h) : 0;     /* kernels.cu:157 */
    if (a || b || e || f) {
        v3 col = lighting(h.mat, s->lights[0].position, s->lights[0].colour, h.hitPoint, c->cam->position, v3_normalise(h.normal), g, px, py);
        if (g->renderHardShadows) {
            if (sphere_shadow(c, &h)) col = v3_scale(col, g->shadowIntensity);
        }
        if (g->reflections && h.objectType == RT_REFLECT && depth < g->maxDepth) {
            v3 r = v3_normalise(v3_reflect(d, h.normal));
            v3 rc = reflection_cast3(c, v3_add(h.hitPoint, v3_scale(h.normal, g->shadowBias)), r, px, py);
            if (h.objectName == RT_PLANE_T) col = v3_add(col, v3_scale(rc, g->planeReflectionStrength));
            else if (h.objectName == RT_SPHERE_T) col = v3_add(col, v3_scale(rc, g->sphereReflectionStrength));
            else if (h.objectName == RT_MODEL_T) col = v3_add(col, v3_scale(rc, g->modelReflectionStrength));
        }
        return col;
    }
    return g->backgroundCol;
}

/* kernels.cu:188-222 */
static v3 reflection_cast1(o_ctx* c, v3 o, v3 d, int px, int py) {
    const rt_scene_desc* s = c->scene; const rt_config* g = c->cfg;
    int depth = 1;
    o_hit h; hit_reset(&h);
    c->rays++;
    int a = trace_spheres(c, o, d, RT_REFLECT_RAY, &h);
    int b = g->reflectPlanes ? trace_planes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int e = g->renderAABBs ? trace_boxes(c, o, d, RT_REFLECT_RAY, &h) : 0;
    int f = g->renderModels ? trace_models(c, o, d, RT_REFLECT_RAY, &h) : 0;
    if (a || b || e || f) {
        v3 col = lighting(h.mat, s->lights[0].position, s->lights[0].colour, h.hitPoint, c->cam->position, v3_normalise(h.normal), g, px, py);
        if (g->reflections && h.objectType == RT_REFLECT && depth < g->maxDepth) {
            v3 r = v3_normalise(v3_reflect(d, h.normal));
            v3 rc = reflection_cast2(c, v3_add(h.hitPoint, v3_scale(h.normal, g->shadowBias)), r, px, py);
            if (h.objectName == RT_PLANE_T) col = v3_add(col, v3_scale(rc, g->planeReflectionStrength));
            else if (h.objectName == RT_SPHERE_T) col = v3_add(col, v3_scale(rc, g->sphereReflectionStrength));
            else if (h.objectName == RT_MODEL_T) col = v3_add(col, v3_scale(rc, g->modelReflectionStrength));
        }
        if (g->renderHardShadows) {
            if (sphere_shadow(c, &h)) col = v3_scale(col, g->shadowIntensity);
        }
        return col;
    }
    return g->backgroundCol;
}

/* kernels.cu:224-318.  *primary receives the closest primary hit (t = 999 on a miss). */
static v3 ray_cast(o_ctx* c, v3 o, v3 d, int px, int py, o_hit* primary) {
    const rt_scene_desc* s = c->scene; const rt_config* g = c->cfg;
    o_hit h; hit_reset(&h);
    c->rays++;
    int a = trace_spheres(c, o, d, RT_PRIMARY_RAY, &h);
    int b = trace_planes(c, o, d, RT_PRIMARY_RAY, &h);
    int e = g->renderAABBs ? trace_boxes(c, o, d, RT_REFLECT_RAY, &h) : 0;        /* kernels.cu:228 */
    int f = g->renderModels ? trace_models(c, o, d, RT_PRIMARY_RAY, &h) : 0;
    if (primary) *primary = h;

    if (a || b || e || f) {
        v3 col;
        if (!h.debug) {
            if (h.objectName == RT_AABB_T)
                col = lighting(h.mat, s->lights[0].position, s->lights[0].colour, h.hitPoint, c->cam->position, h.normal, g, px, py);
            else
                col = lighting(h.mat, s->lights[0].position, s->lights[0].colour, h.hitPoint, c->cam->position, v3_normalise(h.normal), g, px, py);

            if (g->reflections && h.objectType == RT_REFLECT) {
                v3 r = v3_normalise(v3_reflect(d, h.normal));
                if (!(h.objectName == RT_PLANE_T && !g->reflectPlanes)) {
                    v3 rc = reflection_cast1(c, v3_add(h.hitPoint, v3_scale(h.normal, g->shadowBias)), r, px, py);
                    if (h.objectName == RT_PLANE_T) col = v3_add(col, v3_scale(rc, g->planeReflectionStrength));
                    else if (h.objectName == RT_SPHERE_T) col = v3_add(col, v3_scale(rc, g->sphereReflectionStrength));
                    else if (h.objectName == RT_AABB_T) col = v3_add(col, v3_scale(rc, g->AABBReflectionStrength));
                    else if (h.objectName == RT_MODEL_T) col = v3_add(col, v3_scale(rc, g->modelReflectionStrength));
                }
            }

            if (g->renderHardShadows) {                                            /* kernels.cu:268-277 */
                o_hit sh; hit_reset(&sh);
                v3 so = v3_add(h.hitPoint, v3_scale(h.normal, g->shadowBias));
                v3 sd = v3_normalise(v3_sub(s->lights[0].position, h.hitPoint));
                c->rays++;
                int s1 = trace_spheres(c, so, sd, RT_SHADOW_RAY, &sh);
                int s2 = trace_boxes(c, so, sd, RT_SHADOW_RAY, &sh);
                int s3 = trace_models(c, so, sd, RT_SHADOW_RAY, &sh);
                if (sh.t >= 0) {
                    if (s1 || s2 || s3) col = v3_scale(col, g->shadowIntensity);
                }
            }

            if (g->renderSoftShadows) {                                            /* kernels.cu:279-307 */
                int hits = 0;
                for (int i = 0; i < g->softShadowNum; i++) {
                    v3 ps = light_sample_offset(sample_seed(px, py, i), g);
                    v3 lightPoint = v3_add(s->lights[0].position, ps);
                    o_hit sh; hit_reset(&sh);
                    v3 so = v3_add(h.hitPoint, v3_scale(h.normal, g->shadowBias));
                    v3 sd = v3_normalise(v3_sub(lightPoint, h.hitPoint));
                    c->rays++;
                    hits += trace_spheres(c, so, sd, RT_SHADOW_RAY, &sh) ? 1 : 0;
                    hits += trace_boxes(c, so, sd, RT_SHADOW_RAY, &sh) ? 1 : 0;
                    hits += trace_models(c, so, sd, RT_SHADOW_RAY, &sh) ? 1 : 0;
                }
                col = v3_scale(col, 1 - ((float)hits / (float)g->softShadowNum));
            }
        } else {
            col = h.mat.ambientColour;                                             /* kernels.cu:309-312 */
        }
        return col;
    }
    return g->backgroundCol;
}

/* Camera.cpp:22-38 */
static v3 raster_to_camera_space(const rt_camera* cam, float x, float y, int width, int height) {
    float ndcx = x / (float)width, ndcy = y / (float)height;
    float sx = 2 * ndcx - 1, sy = 1 - 2 * ndcy;
    float cx = sx * 1 * cam->fovFactor, cy = sy * cam->fovFactor, cz = 1;
    return v3_add(v3_add(v3_scale(cam->right, cx), v3_scale(cam->up, cy)), v3_scale(cam->direction, cz));
}

/* kernels.cu:361-363 — min(c, 1) * 255 stored to a byte: truncation, negatives and NaN products -> 0 */
static uint8_t pack_channel(float cval) {
    float v = f_min(cval, 1.0f) * 255;
    if (!(v > 0)) return 0;
    return (uint8_t)f2i_rz(v);
}

/* kernels.cu:320-364 for one pixel */
static void render_pixel(o_ctx* c, int x, int y, int width, int height, uint8_t* fb, int32_t* ids, float* tbuf) {
    const rt_config* g = c->cfg;
    v3 col;
    o_hit primary; hit_reset(&primary);
    if (g->antiAliasing) {
        v3 total = v3_make(0, 0, 0);
        int first = 1;
        for (float i = 0.25f; i <= 0.75f; i += 0.5f) {
            for (float j = 0.25f; j <= 0.75f; j += 0.5f) {
                v3 p = raster_to_camera_space(c->cam, (float)(x + i), (float)(y + j), width, height);
                o_hit ph;
                total = v3_add(total, ray_cast(c, c->cam->position, v3_normalise(p), x, y, &ph));
                if (first) { primary = ph; first = 0; }
            }
        }
        col = v3_divs(total, 4.0f);
    } else {
        v3 p = raster_to_camera_space(c->cam, (float)(x + 0.5), (float)(y + 0.5), width, height);
        col = ray_cast(c, c->cam->position, v3_normalise(p), x, y, &primary);
    }
    size_t pix = (size_t)y * (size_t)width + (size_t)x;
    fb[pix * 3 + 0] = pack_channel(col.x);
    fb[pix * 3 + 1] = pack_channel(col.y);
    fb[pix * 3 + 2] = pack_channel(col.z);
    if (ids) {
        ids[pix * 2 + 0] = (primary.t != FAR_T) ? (primary.objectName << 24) | (primary.index & 0xFFFFFF) : -1;
        ids[pix * 2 + 1] = (primary.t != FAR_T) ? primary.tri : -1;
    }
    if (tbuf) tbuf[pix] = primary.t;
}

/* ---------------------------------------------------------- entry points */

/*
 * Renders pixels [x0,x1) x [y0,y1) of a width x height frame (kernels.cu:320-364
 * over that window) into fb (3 bytes per pixel, row-major over the FULL frame).
 * ids (optional): two int32 per pixel, (objectName << 24 | index, triangle) or (-1, -1).
 * tbuf (optional): primary hit distance (999 = miss).  counters (optional): [0] rays, [1] triangle tests.
 */
void oracle_render(const rt_scene_desc* scene, const rt_camera* cam, const rt_config* cfg,
                   int width, int height, int x0, int y0, int x1, int y1,
                   uint8_t* fb, int32_t* ids, float* tbuf, uint64_t* counters, int nthreads) {
    uint64_t rays = 0, tris = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : rays, tris)
#endif
    for (int y = y0; y < y1; y++) {
        o_ctx c; c.scene = scene; c.cam = cam; c.cfg = cfg; c.rays = 0; c.tri_tests = 0;
        for (int x = x0; x < x1; x++) render_pixel(&c, x, y, width, height, fb, ids, tbuf);
        rays += c.rays; tris += c.tri_tests;
    }
    if (counters) { counters[0] = rays; counters[1] = tris; }
}

/* AABB.cpp:26-34 — the subtraction of 0.1 happens in double, then narrows */
static void aabb_extend_by(v3* bmin, v3* bmax, v3 p) {
    if (p.x < bmin->x) bmin->x = (float)((double)p.x - 0.1);
    if (p.y < bmin->y) bmin->y = (float)((double)p.y - 0.1);
    if (p.z < bmin->z) bmin->z = (float)((double)p.z - 0.1);
    if (p.x > bmax->x) bmax->x = (float)((double)p.x + 0.1);
    if (p.y > bmax->y) bmax->y = (float)((double)p.y + 0.1);
    if (p.z > bmax->z) bmax->z = (float)((double)p.z + 0.1);
}

/*
 * Model.cpp:29-129 — bakes loader output into world space and builds the bounding volumes.
 * Inputs (per loaded vertex, in LoadedVertices order): pos[3n], nrm[3n], tex[2n], ka/kd/ks[3n] of the
 * vertex's mesh material; loadedIndices[indicesCount].
 * Outputs: vertices[n]; indices[indicesCount] (the final triple is left 0, Model.cpp:70);
 * bb[6] = whole-model box min,max; chunkBoxes[8*6]; chunkIdx = 8 arrays of *indicesPerBox ints,
 * caller-allocated with at least 3*ceil(ceil(indicesCount/3)/8) entries each, zero-filled here first
 * (the reference leaves unfilled slots of its freshly allocated arrays untouched; they are read as 0).
 */
void oracle_bake_model(const float* pos, const float* nrm, const float* tex,
                       const float* ka, const float* kd, const float* ks, int vertCount,
                       const uint32_t* loadedIndices, int indicesCount,
                       const float* position, int size,
                       rt_vertex* vertices, int32_t* indices, float* bb, float* chunkBoxes,
                       int32_t* chunkIdx, int32_t* indicesPerBoxOut) {
    int trianglesPerBox = (int)ceil(ceil((double)(indicesCount / 3)) / 8);      /* Model.cpp:62 */
    int indicesPerBox = trianglesPerBox * 3;
    *indicesPerBoxOut = indicesPerBox;
    memset(indices, 0, sizeof(int32_t) * (size_t)indicesCount);
    memset(chunkIdx, 0, sizeof(int32_t) * (size_t)indicesPerBox * 8);

    int box = 0, slot = 0;
    for (int i = 0; i < indicesCount - 3; i += 3) {                              /* Model.cpp:70-84 */
        for (int k = 0; k < 3; k++) {
            indices[i + k] = (int32_t)loadedIndices[i + k];
            chunkIdx[(size_t)box * indicesPerBox + slot + k] = (int32_t)loadedIndices[i + k];
        }
        slot += 3;
        if (slot >= indicesPerBox) { box++; slot = 0; }
    }

    v3 bmin = v3_make(0, 0, 0), bmax = v3_make(0, 0, 0);                          /* AABB.cpp:3-6 */
    v3 P = v3_make(position[0], position[1], position[2]);
    for (int j = 0; j < vertCount; j++) {                                         /* Model.cpp:88-117 */
        rt_vertex* v = &vertices[j];
        memset(v, 0, sizeof *v);
        v->position = v3_make(P.x + pos[3 * j] * (float)size, P.y + -pos[3 * j + 1] * (float)size, P.z + pos[3 * j + 2] * (float)size);
        v->textureCoords = v3_make(tex[2 * j], tex[2 * j + 1], 0);
        v->normal = v3_make(nrm[3 * j], -nrm[3 * j + 1], nrm[3 * j + 2]);
        aabb_extend_by(&bmin, &bmax, v->position);
        v->ambient = v3_make(ka[3 * j], ka[3 * j + 1], ka[3 * j + 2]);
        v->diffuse = v3_make(kd[3 * j], kd[3 * j + 1], kd[3 * j + 2]);
        v->specular = v3_make(ks[3 * j], ks[3 * j + 1], ks[3 * j + 2]);
    }
    bb[0] = bmin.x; bb[1] = bmin.y; bb[2] = bmin.z; bb[3] = bmax.x; bb[4] = bmax.y; bb[5] = bmax.z;

    for (int b = 0; b < 8; b++) {                                                 /* Model.cpp:21-27 */
        v3 cmin = v3_make(0, 0, 0), cmax = v3_make(0, 0, 0);
        for (int j = 0; j < indicesPerBox; j++)
            aabb_extend_by(&cmin, &cmax, vertices[chunkIdx[(size_t)b * indicesPerBox + j]].position);
        float* o = chunkBoxes + 6 * b;
        o[0] = cmin.x; o[1] = cmin.y; o[2] = cmin.z; o[3] = cmax.x; o[4] = cmax.y; o[5] = cmax.z;
    }
}

/* Known-answer entry points for the primitive routines (one call = one reference routine). */
int oracle_solve_quadratic(float a, float b, float c, float* x0, float* x1) { return solve_quadratic(a, b, c, x0, x1); }
float oracle_wang_hash(uint32_t seed) { return wang_hash(seed); }
int oracle_sphere_hit(const rt_sphere* s, const float* o, const float* d, float* t0, float* t1) {
    return sphere_hit(s, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t0, t1);
}
int oracle_plane_hit(const rt_plane* p, const float* o, const float* d, float* t0) {
    return plane_hit(p, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t0);
}
int oracle_aabb_hit(const float* bmin, const float* bmax, const float* o, const float* d, float* t0, float* t1) {
    return aabb_hit(v3_make(bmin[0], bmin[1], bmin[2]), v3_make(bmax[0], bmax[1], bmax[2]),
                    v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t0, t1);
}
void oracle_aabb_normal_at(const float* bmin, const float* bmax, const float* p, float* out) {
    v3 n = aabb_normal_at(v3_make(bmin[0], bmin[1], bmin[2]), v3_make(bmax[0], bmax[1], bmax[2]), v3_make(p[0], p[1], p[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
int oracle_triangle_intersect(const float* p0, const float* p1, const float* p2, const float* o, const float* d,
                              int cull, float* t, float* u, float* v) {
    return triangle_intersect(v3_make(p0[0], p0[1], p0[2]), v3_make(p1[0], p1[1], p1[2]), v3_make(p2[0], p2[1], p2[2]),
                              v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t, u, v, cull);
}
int oracle_triangle_object_hit(const float* a, const float* b, const float* c, const float* o, const float* d, float* t0) {
    return triangle_object_hit(v3_make(a[0], a[1], a[2]), v3_make(b[0], b[1], b[2]), v3_make(c[0], c[1], c[2]),
                               v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]), t0);
}
void oracle_camera_ray(const rt_camera* cam, float x, float y, int width, int height, float* out) {
    v3 p = v3_normalise(raster_to_camera_space(cam, x, y, width, height));
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void oracle_lighting(const rt_material* mat, const float* lightPos, const float* lightCol, const float* point,
                     const float* eye, const float* normal, const rt_config* cfg, int px, int py, float* out) {
    v3 c = lighting(*mat, v3_make(lightPos[0], lightPos[1], lightPos[2]), v3_make(lightCol[0], lightCol[1], lightCol[2]),
                    v3_make(point[0], point[1], point[2]), v3_make(eye[0], eye[1], eye[2]),
                    v3_make(normal[0], normal[1], normal[2]), cfg, px, py);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
uint8_t oracle_pack_channel(float c) { return pack_channel(c); }

/* Camera.cpp:54-63 with utils.h:102-104 — basis from yaw/pitch in degrees; Camera.cpp:10 — fovFactor. */
void oracle_camera_basis(const float* position, float fov, float yaw, float pitch, rt_camera* out) {
    float pitchR = (float)((double)pitch * 0.01745329251994329576923690768489);
    float yawR = (float)((double)yaw * 0.01745329251994329576923690768489);
    v3 dir = v3_normalise(v3_make(cosf(yawR) * cosf(pitchR), sinf(pitchR), sinf(yawR) * cosf(pitchR)));
    v3 right = v3_normalise(v3_cross(dir, v3_make(0.0f, 1.0f, 0.0f)));
    v3 up = v3_normalise(v3_cross(right, dir));
    out->position = v3_make(position[0], position[1], position[2]);
    out->direction = dir; out->right = right; out->up = up;
    out->fovFactor = tanf((float)((double)(fov / 2) * 0.01745329251994329576923690768489));
}

/* utils.h:102-104 — the factor is a double literal, the result narrows to float */
static float radians_f(float angle) { return (float)((double)angle * 0.01745329251994329576923690768489); }

/*
 * Batched forms of the known-answer entry points, n independent cases per call (vectors as 3 floats per case), in the
 * shapes of oracle/ref_driver.cpp so that the same arrays can go through this restatement and through the reference's own
 * compiled routines.  Outputs that a routine does not write keep their incoming values.
 */
static v3 v3_at(const float* p, int i) { return v3_make(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
static void v3_put(float* p, int i, v3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

void oracle_kat_solve_quadratic(int n, const float* a, const float* b, const float* c, int* ok, float* x0, float* x1) {
    for (int i = 0; i < n; i++) ok[i] = solve_quadratic(a[i], b[i], c[i], &x0[i], &x1[i]);
}
void oracle_kat_sphere_hit(int n, const float* centre, const float* radius, const float* o, const float* d, int* ok, float* t0, float* t1) {
    for (int i = 0; i < n; i++) {
        rt_sphere s; memset(&s, 0, sizeof s);
        s.position = v3_at(centre, i); s.radius = radius[i];
        ok[i] = sphere_hit(&s, v3_at(o, i), v3_at(d, i), &t0[i], &t1[i]);
    }
}
void oracle_kat_plane_hit(int n, const float* p, const float* normal, const float* o, const float* d, int* ok, float* t0) {
    for (int i = 0; i < n; i++) {
        rt_plane pl; memset(&pl, 0, sizeof pl);
        pl.position = v3_at(p, i); pl.n = v3_at(normal, i);
        ok[i] = plane_hit(&pl, v3_at(o, i), v3_at(d, i), &t0[i]);
    }
}
void oracle_kat_triangle_object_hit(int n, const float* a, const float* b, const float* c, const float* o, const float* d, int* ok, float* t0) {
    for (int i = 0; i < n; i++) ok[i] = triangle_object_hit(v3_at(a, i), v3_at(b, i), v3_at(c, i), v3_at(o, i), v3_at(d, i), &t0[i]);
}
void oracle_kat_vec3_ops(int n, const float* a, const float* b, float* length, float* dotOut, float* crossOut, float* normalised,
                         float* reflected, float* rad) {
    for (int i = 0; i < n; i++) {
        v3 x = v3_at(a, i), y = v3_at(b, i);
        length[i] = v3_length(x);
        dotOut[i] = v3_dot(x, y);
        v3_put(crossOut, i, v3_cross(x, y));
        v3_put(normalised, i, v3_normalise(x));
        v3_put(reflected, i, v3_reflect(x, y));
        rad[i] = radians_f(a[3 * i]);
    }
}
/* kernels.cu:337-338, 355-356: raw = rasterToCameraSpace(float(x + off), float(y + off), w, h), dir = normalise(raw) */
void oracle_kat_camera_rays(const rt_camera* cam, int width, int height, int x0, int y0, int x1, int y1, double offX, double offY,
                            float* raw, float* dir) {
    int k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++, k++) {
            v3 p = raster_to_camera_space(cam, (float)(x + offX), (float)(y + offY), width, height);
            if (raw) v3_put(raw, k, p);
            if (dir) v3_put(dir, k, v3_normalise(p));
        }
}
/*
 * Model::hit without a box (Model.cpp:318-356) for nRays rays on one mesh: vertices as 12 floats (position, normal,
 * ambient, diffuse).  hit / t0 / attrs (interpolated normal, ambient, diffuse: 9 floats) per ray.
 */
void oracle_kat_model_hit_nobox(int nVerts, const float* verts12, int nIndices, const int32_t* indices, int nRays, const float* rays6,
                                int cull, int* hit, float* t0, float* attrs9) {
    rt_vertex* vertices = (rt_vertex*)calloc((size_t)nVerts + 1, sizeof(rt_vertex));
    for (int i = 0; i < nVerts; i++) {
        const float* f = verts12 + 12 * (size_t)i;
        vertices[i].position = v3_make(f[0], f[1], f[2]); vertices[i].normal = v3_make(f[3], f[4], f[5]);
        vertices[i].ambient = v3_make(f[6], f[7], f[8]); vertices[i].diffuse = v3_make(f[9], f[10], f[11]);
    }
    rt_model m; memset(&m, 0, sizeof m);
    m.vertices = vertices; m.vertCount = nVerts; m.indices = indices; m.indicesCount = nIndices;
    for (int r = 0; r < nRays; r++) {
        const float* q = rays6 + 6 * (size_t)r;
        hit_vertex hv; memset(&hv, 0, sizeof hv);
        hit[r] = model_hit(&m, v3_make(q[0], q[1], q[2]), v3_make(q[3], q[4], q[5]), &t0[r], &hv, 0, 0, 0, cull, NULL);
        float* a = attrs9 + 9 * (size_t)r;
        a[0] = hv.normal.x; a[1] = hv.normal.y; a[2] = hv.normal.z; a[3] = hv.ambient.x; a[4] = hv.ambient.y; a[5] = hv.ambient.z;
        a[6] = hv.diffuse.x; a[7] = hv.diffuse.y; a[8] = hv.diffuse.z;
    }
    free(vertices);
}
/*
 * One single-triangle model per case (15 floats: v0, v1, v2, origin, direction), as oracle/ref_model_driver.cpp's "tri" mode
 * builds them: the index list names the triangle twice (the loop stops before the last one, Model.cpp:324) and the vertex
 * normals are the unit axes, so the interpolated normal is (u, v, 1-u-v).
 */
void oracle_kat_triangle_cases(int nCases, const float* cases15, int cull, int* hit, float* t0, float* attrs9) {
    rt_vertex vertices[3];
    memset(vertices, 0, sizeof vertices);
    const int32_t indices[6] = { 0, 1, 2, 0, 1, 2 };
    vertices[0].normal = v3_make(1, 0, 0); vertices[1].normal = v3_make(0, 1, 0); vertices[2].normal = v3_make(0, 0, 1);
    rt_model m; memset(&m, 0, sizeof m);
    m.vertices = vertices; m.vertCount = 3; m.indices = indices; m.indicesCount = 6;
    for (int c = 0; c < nCases; c++) {
        const float* q = cases15 + 15 * (size_t)c;
        for (int k = 0; k < 3; k++) vertices[k].position = v3_make(q[3 * k], q[3 * k + 1], q[3 * k + 2]);
        hit_vertex hv; memset(&hv, 0, sizeof hv);
        hit[c] = model_hit(&m, v3_make(q[9], q[10], q[11]), v3_make(q[12], q[13], q[14]), &t0[c], &hv, 0, 0, 0, cull, NULL);
        float* a = attrs9 + 9 * (size_t)c;
        a[0] = hv.normal.x; a[1] = hv.normal.y; a[2] = hv.normal.z; a[3] = hv.ambient.x; a[4] = hv.ambient.y; a[5] = hv.ambient.z;
        a[6] = hv.diffuse.x; a[7] = hv.diffuse.y; a[8] = hv.diffuse.z;
    }
}
