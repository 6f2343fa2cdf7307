// gfx950 render kernels: one wave per 8x8 pixel block, one lane per pixel; the mesh traversal is a wave-wide operation in
// which the lanes share the work of all 64 rays (tracePool).
//
// Behavioural contract (what each routine must reproduce, reference file:line):
//   frame loop / pixel pack      kernels.cu:320-364        camera ray          Camera.cpp:22-38
//   primary cast + shadows       kernels.cu:224-318        reflection chain    kernels.cu:130-222
//   linear object traces         kernels.cu:67-96          model trace         kernels.cu:98-128
//   sphere / quadratic           Sphere.cpp:17-48, utils.cpp:10-32
//   plane                        Plane.cpp:16-26           box + box normal    AABB.cpp:36-95
//   triangle test                Model.cpp:135-159         mesh hit (box gate + all triangles)  Model.cpp:275-317
//   Phong lighting, hash         kernels.cu:11-65
//
// Arithmetic: this file is compiled with -ffp-contract=off and correctly rounded division / square root, and
// every value that decides a hit or a colour is computed with the reference's operations in the reference's
// order, so results agree bit for bit with a plain IEEE evaluation on the host (libm calls excepted).  The only
// approximate arithmetic is the octree child-box slab test (explicit fma + reciprocal), which is conservative:
// child boxes are padded on the host and the test never rejects a box containing an acceptable triangle.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <mutex>

#include "rt_device_types.h"
#include "rt_kernels.h"
#include "rt_math.h"

namespace rt {

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

#define RT_DEV __device__ __forceinline__
constexpr int kBlockThreads = 64;   // render workgroups are single waves
constexpr uint32_t kSplitGainNum = 5, kSplitGainDen = 3;   // a block rendered by eight split waves is taken to finish 5/3 times sooner (measured: 1.65)
// Occupancy of the render kernel: four waves per SIMD (128 registers; what does not fit is spilled around the traces, not
// inside them) with 10 KB of LDS per wave - measured 5 % faster than three waves with 168 registers and 12.4 KB.
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4
#endif
#ifndef RT_NODE_POOL_ENTRIES
#define RT_NODE_POOL_ENTRIES 512
#endif
#ifndef RT_RCP_IN_LDS
#define RT_RCP_IN_LDS 1
#endif
constexpr int kNodePool = RT_NODE_POOL_ENTRIES;       // mesh traversal: entries of pending children per wave
#ifndef RT_TRI_POOL_ENTRIES
#define RT_TRI_POOL_ENTRIES 288
#endif
constexpr int kTriPool = RT_TRI_POOL_ENTRIES;        // and of pending triangle work: runs of triangles, and sets of leaves of one node
constexpr uint32_t kNodeIndexBits = 29, kNodeIndexMask = (1u << kNodeIndexBits) - 1u;   // node pool entry, low word: first child | octant << 29
constexpr uint32_t kLeafSet = 1u << 27;   // triangle pool entry, high word: owner | chunks << 8 | (leaf set: hit mask << 16 | octant << 24 | kLeafSet)
// Triangle turns are taken once this many entries are pending (1080p benchmark frame: 24 -> +8 %, 40 -> +2 %, 52 .. 58 within
// 0.5 % of each other, 64 -> +1 %, 96 -> +4 %).
#ifndef RT_TRI_TURN_AT
#define RT_TRI_TURN_AT 56u
#endif
constexpr int kDirectStartModels = 6;  // up to this many models a trace can start at the model roots (one pool entry per ray and model)
constexpr int kExactRankTiles = 4096;  // up to this many tiles are ranked by all-pairs comparison
constexpr int kWaveRecordWords = 18; // per-wave diagnostic record: start, end, 8 x (cycles, executions)
#define FAR_T 999.0f

struct V3 { float x, y, z; };

RT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 mk(const float* p) { return mk(p[0], p[1], p[2]); }
// Values that are the same for the whole launch and lie in the kernel-argument segment (or, for a camera path, in a table in
// global memory that nothing writes while the kernel runs) are read where they are used, through the constant address space:
// scalar loads served by the scalar cache.  Loaded once at kernel entry instead, the hundred-odd words would have to live in
// scalar registers for the whole kernel - there are 102 - and most of them would be moved into spare vector-register lanes
// and back (one vector instruction each way).
// The accesses are volatile ones: each read in the source is one scalar load at that place - the compiler neither moves it to
// the top of the kernel (from where the value would have to be kept) nor merges it with another read, so code that needs a
// value several times reads it into a local variable first.
#define RT_KARG __attribute__((address_space(4)))
typedef const volatile RT_KARG FrameParams FrameK;
typedef const volatile RT_KARG DevScene SceneK;
typedef const volatile RT_KARG float* CamK;      // 16 floats: position, up, right, direction, fovFactor, view vector (FrameParams::camPos .. eyeDir)
RT_DEV V3 mk(const volatile RT_KARG float* p) { return mk(p[0], p[1], p[2]); }
// A pointer read from the argument segment is a generic one; through this cast the accesses become global ones.
template <typename T> RT_DEV T* inGlobal(T* p) { return (T*)(__attribute__((address_space(1))) T*)p; }
RT_DEV V3 xyz(f4_t v) { return mk(v.x, v.y, v.z); }
RT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float n) { return mk(a.x * n, a.y * n, a.z * n); }
RT_DEV V3 operator/(V3 a, float n) { return mk(a.x / n, a.y / n, a.z / n); }
RT_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_DEV V3 normalise(V3 a) { return a / length(a); }
RT_DEV V3 reflect(V3 a, V3 b) { return a - (b * dot(a, b)) * mk(2.0f, 2.0f, 2.0f); }
RT_DEV float selMax(float x, float y) { return (x > y) ? x : y; }   // utils.h:91-93
RT_DEV float selMin(float x, float y) { return (x < y) ? x : y; }   // utils.h:86-88

RT_DEV f4_t ld4(const F4* p) { return *reinterpret_cast<const f4_t*>(p); }
typedef float f3_t __attribute__((ext_vector_type(3)));
// The first three words of a record whose fourth is not needed: a 12-byte request, and no register that waits for a word
// nobody reads (the code after the loads would reuse such a register at once, and wait for the load to do so).
RT_DEV f3_t ld3(const F4* p) { return *reinterpret_cast<const f3_t*>(p); }
// Loads at an address that is the same for the whole wave, from tables nothing writes while a render kernel runs (the scene's
// object tables): through the constant address space, so that the scalar unit fetches them (its cache holds these few hundred
// bytes for good, and the values arrive in scalar registers) instead of 64 lanes asking the vector cache for one address.
RT_DEV f4_t ldu4(const F4* p) { return *(const __attribute__((address_space(4))) f4_t*)(unsigned long long)p; }
RT_DEV uint32_t lduw(const uint32_t* p) { return *(const __attribute__((address_space(4))) uint32_t*)(unsigned long long)p; }
RT_DEV uint32_t lduFlags(const F4* p) { return lduw(reinterpret_cast<const uint32_t*>(p) + 3); }     // the w word of a table entry
RT_DEV uint32_t fbits(float f) { return __float_as_uint(f); }
RT_DEV float ubits(uint32_t u) { return __uint_as_float(u); }

// float -> int, toward zero, saturating, NaN -> 0 (v_cvt_i32_f32 semantics, spelled out).
RT_DEV int f2iRz(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// ---------------------------------------------------------------- primitives (exact)

RT_DEV bool solveQuadratic(float a, float b, float c, float& x0, float& x1) {
    float disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    if (disc == 0) {
        x0 = x1 = (float)(-0.5 * (double)b / (double)a);
    } else {
        float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(disc))) : (float)(-0.5 * (double)(b - sqrtf(disc)));
        x0 = q / a;
        x1 = c / q;
    }
    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
    return true;
}

RT_DEV bool sphereHit(V3 centre, float radius, V3 o, V3 d, float& t0) {
    float a0, a1;
    V3 L = o - centre;
    float a = dot(d, d);
    float b = 2 * dot(d, L);
    float c = dot(L, L) - radius * radius;
    if (!solveQuadratic(a, b, c, a0, a1)) return false;
    if (a0 < 0) { a0 = a1; if (a0 < 0) return false; }
    if (a1 < 0) { a1 = a0; if (a1 < 0) return false; }
    if (a0 > a1) { float t = a1; a1 = a0; a0 = t; }
    t0 = a0;
    return true;
}

RT_DEV bool planeHit(V3 p, V3 n, V3 o, V3 d, float& t0) {
    float denom = dot(n, d);
    V3 v = p - o;
    t0 = dot(v, n) / denom;
    return t0 >= 0;
}

// The reference's slab test, branch for branch (its NaN behaviour decides whether a mesh is tested at all).
RT_DEV bool boxHit(V3 bmin, V3 bmax, V3 o, V3 d, float& t0) {
    float tmp;
    float tmin = (bmin.x - o.x) / d.x, tmax = (bmax.x - o.x) / d.x;
    if (tmin > tmax) { tmp = tmin; tmin = tmax; tmax = tmp; }
    float tminY = (bmin.y - o.y) / d.y, tmaxY = (bmax.y - o.y) / d.y;
    if (tminY > tmaxY) { tmp = tminY; tminY = tmaxY; tmaxY = tmp; }
    if ((tmin > tmaxY) || (tminY > tmax)) return false;
    if (tminY > tmin) tmin = tminY;
    if (tmaxY < tmax) tmax = tmaxY;
    float tminZ = (bmin.z - o.z) / d.z, tmaxZ = (bmax.z - o.z) / d.z;
    if (tminZ > tmaxZ) { tmp = tminZ; tminZ = tmaxZ; tmaxZ = tmp; }
    if ((tmin > tmaxZ) || (tminZ > tmax)) return false;
    if (tminZ > tmin) tmin = tminZ;
    if (tmaxZ < tmax) tmax = tmaxZ;
    if (tmin < 0) { tmin = tmax; if (tmin < 0) return false; }
    if (tmax < 0) { tmax = tmin; if (tmax < 0) return false; }
    if (tmin > tmax) { tmp = tmin; tmin = tmax; tmax = tmp; }
    t0 = tmin;
    return true;
}

// Whether boxHit accepts the ray, for callers that do not need the distance.  The six quotients are first formed with
// the ray's reciprocals (v_rcp_f32, within 2^-22 of the correctly rounded division relative to the quotient); when
// the slab intervals overlap, or fail to, by more than four times that uncertainty the outcome of boxHit follows
// without a division.  Everything else - a near-tangent ray, a direction component that is zero or denormal, a box
// touching the origin - goes through boxHit itself.  rdFinite: all three reciprocals are finite.
RT_DEV bool boxHitDecision(V3 bmin, V3 bmax, V3 o, V3 d, V3 rd, bool rdFinite) {
    const float ax = (bmin.x - o.x) * rd.x, bx = (bmax.x - o.x) * rd.x;
    const float ay = (bmin.y - o.y) * rd.y, by = (bmax.y - o.y) * rd.y;
    const float az = (bmin.z - o.z) * rd.z, bz = (bmax.z - o.z) * rd.z;
    const float enter = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float leave = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float big = fmaxf(fmaxf(fmaxf(fabsf(ax), fabsf(bx)), fmaxf(fabsf(ay), fabsf(by))), fmaxf(fabsf(az), fabsf(bz)));
    const float margin = big * 1e-6f;
    const bool sane = rdFinite && big > 1e-30f && big < 1e30f;
    if (sane && leave - enter > 2 * margin && leave > margin) return true;
    if (sane && (enter - leave > 2 * margin || leave < -margin)) return false;
    float t0;
    return boxHit(bmin, bmax, o, d, t0);
}

RT_DEV V3 boxNormalAt(V3 bmin, V3 bmax, V3 point) {
    V3 centre = (bmin + bmax) / 2.0f;
    V3 p = point - centre;
    V3 e = bmin - bmax;
    V3 h = mk(fabsf(e.x), fabsf(e.y), fabsf(e.z)) / 2.0f;
    const float bias = 1.00001f;
    V3 n = mk((float)f2iRz(p.x / h.x * bias), (float)f2iRz(p.y / h.y * bias), (float)f2iRz(p.z / h.z * bias));
    return normalise(n);
}

RT_DEV float wangHash(uint32_t seed) {
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return (float)seed / 4294967296.0f;
}

RT_DEV uint32_t sampleSeed(int x, int y, int i) { return ((uint32_t)x * 1000u + (uint32_t)y) * 1973u + (uint32_t)i * 9277u; }

// A point on the light's sphere from two hashed numbers (kernels.cu:44-46, 295-297).  The reference calls sin / cos / acos
// on floats; every float math library rounds those a little differently, and a shadow ray whose direction moves by one
// unit in the last place can flip at an occluder's edge.  Here (and in the CPU oracle) the three functions are evaluated in
// double and rounded to float once, which gives the correctly rounded float result - the one value two independent
// libraries agree on.  One out-of-line copy: the double routines are long, and soft shadows are not on the benchmark's path.
__device__ __noinline__ f4_t lightSampleAngles(float theta, float cosPhi) {
    const double phi = (double)(float)acos((double)cosPhi);       // float phi = acos(...)
    f4_t r = {(float)sin(phi), (float)cos(phi), (float)sin((double)theta), (float)cos((double)theta)};
    return r;
}

RT_DEV V3 lightSampleOffset(uint32_t seed, float radius) {
    float theta = 2 * 3.141592654f * wangHash(seed);
    const f4_t a = lightSampleAngles(theta, 2 * wangHash(seed * 16807u) - 1);
    const float sinPhi = a.x, cosPhi = a.y, sinTheta = a.z, cosTheta = a.w;
    return mk(radius * sinPhi * cosTheta, radius * sinPhi * sinTheta, radius * cosPhi);
}

// ---------------------------------------------------------------- per-ray working set

struct Mat { V3 ka, kd; float ambient, diffuse, specular, shininess; uint32_t type, flags; };

RT_DEV Mat loadMat(const F4* base, int idx) {
    f4_t a = ld4(base + 3 * idx), b = ld4(base + 3 * idx + 1), c = ld4(base + 3 * idx + 2);
    Mat m;
    m.ka = xyz(a); m.ambient = a.w; m.kd = xyz(b); m.diffuse = b.w;
    m.specular = c.x; m.shininess = c.y; m.type = fbits(c.z); m.flags = fbits(c.w);
    return m;
}

// A model's material without its two colours, which the hit's interpolated vertex colours replace (kernels.cu:114-117): only
// the words that are read are requested, so that no register waits for a colour nobody looks at (it would be handed to the
// address of the attribute fetch that follows, and that fetch would wait for the material).
RT_DEV Mat loadMatWithoutColours(const F4* base, int idx) {
    const float* w = reinterpret_cast<const float*>(base + 3 * idx);
    const float ambient = w[3], diffuse = w[7];
    f4_t c = ld4(base + 3 * idx + 2);
    Mat m;
    m.ka = mk(0, 0, 0); m.ambient = ambient; m.kd = mk(0, 0, 0); m.diffuse = diffuse;
    m.specular = c.x; m.shininess = c.y; m.type = fbits(c.z); m.flags = fbits(c.w);
    return m;
}

enum { KIND_NONE = -1, KIND_SPHERE = 0, KIND_PLANE = 1, KIND_MODEL = 4, KIND_BOX = 5 };   // utils.h:14-16 values

struct Closest {
    float t;      // FAR_T = nothing yet
    int kind, idx;
    int tri;      // model hits: the triangle's position in the scene-wide attribute array (the model's base + its index in the model)
    float u, v;
};

struct Counters { uint32_t rays, nodeTests, triTests, attrFetches, gateTests, nodeVisits, tableFlops, usefulNodeTests, usefulTriTests, usefulNodeVisits; };

// The wave's working set in LDS for the mesh traversal (tracePool); everything is indexed by lane or by pool position.
struct WaveLds {
    f4_t* rayO; f4_t* rayD; f4_t* rayR;  // [64] ray of lane l: origin | gate bits low, direction | gate bits high, reciprocal direction | octant
    unsigned long long* bestKey;         // [64] ordered hit distance << 32 | triangle order key (see tracePool)
    uint32_t* bestRec;                   // [64] triangle record of the hit in bestKey
    unsigned long long* nodePool;        // [kNodePool] pending children: node base | ray octant << 29 | (imask | hits << 8 | owner << 16 | chunks << 24) << 32
    unsigned long long* triPool;         // [kTriPool] pending triangles: first record | (owner | chunks << 8) << 32, or a node's leaf set (kLeafSet)
    unsigned long long* prof;   // instrumented kernel only: 8 per-wave accumulators in LDS (see renderKernel)
};

// Wave-level section timer for the instrumented kernel: the first active lane adds the elapsed shader cycles and one
// iteration to the wave's accumulators.
template <bool STATS>
struct SectionTimer {
    unsigned long long t0;
    RT_DEV SectionTimer() : t0(STATS ? __builtin_readcyclecounter() : 0ull) {}
    RT_DEV void stop(const WaveLds& lds, int slot) {
        if (STATS) {
            unsigned long long dt = __builtin_readcyclecounter() - t0;
            unsigned long long active = __ballot(1);
            if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)active) - 1)) { lds.prof[2 * slot] += dt; lds.prof[2 * slot + 1] += 1ull + ((unsigned long long)__builtin_popcountll(active) << 32); }
        }
    }
};

// ---------------------------------------------------------------- octree traversal

// Reorders the 8 child bits of each byte of m so that bit k is child k ^ oct (front-to-back for a ray whose negative axes
// are oct).  All four bytes are permuted alike, so several masks go through in one call.
RT_DEV uint32_t permuteOctants(uint32_t m, uint32_t oct) {
    if (oct & 1u) m = ((m & 0x55555555u) << 1) | ((m & 0xAAAAAAAAu) >> 1);
    if (oct & 2u) m = ((m & 0x33333333u) << 2) | ((m & 0xCCCCCCCCu) >> 2);
    if (oct & 4u) m = ((m & 0x0F0F0F0Fu) << 4) | ((m & 0xF0F0F0F0u) >> 4);
    return m;
}

typedef float f2_t __attribute__((ext_vector_type(2)));

// Per-ray constants of the approximate slab tests: rd = 1/d (v_rcp_f32, 1 ulp) and the origin.  A plane at coordinate p
// lies at ray parameter (p - o) * rd: the subtraction comes first, so the rounding error is relative to the distance between
// the ray origin and the plane, wherever in the world the two are.
struct SlabRay {
    V3 rd, o;
    uint32_t oct;            // bit a set: the ray travels towards -a (sign bit of d, so that -0 pairs with rd = -inf)
};

RT_DEV SlabRay makeSlabRay(V3 o, V3 d) {
    SlabRay r;
    r.rd = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    r.o = o;
    r.oct = (fbits(d.x) >> 31) | ((fbits(d.y) >> 31) << 1) | ((fbits(d.z) >> 31) << 2);
    return r;
}

// A wave raises its issue priority from the start of a traversal turn until its node or triangle loads are out, and drops it
// for the arithmetic that follows: of the four waves on a SIMD, the one that is about to wait for memory goes first, so its
// latency runs under the others' arithmetic (measured: frame 1 % shorter on a latency-bound scene, 2.5-3 % on issue-bound ones;
// any level above 0 does the same, holding it until the ray has arrived from LDS as well costs 3 %).
// A turn ends with none of its loads outstanding.  They have long arrived by then; but a turn that rejected all its triangles
// at the first test, or whose entries all belonged to settled rays, never waited for the records it had requested, and the
// compiler, unable to tell, makes the next turn wait for ALL loads - its own fresh ones included - before the first write to a
// register such a record was bound for (measured: the second LDS read of the ray sat behind the node loads instead of under them).
RT_DEV void loadsHaveArrived() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), the other counters left alone

RT_DEV void fetchPriority(bool raised) {
    if (raised) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
}

// Sum of v over the 64 lanes of the wave, all of which must be active: four shifts within the rows of sixteen lanes, then the
// row totals are passed on to the next row and to the upper half; lane 63 ends up with the total.
RT_DEV uint32_t waveSum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

RT_DEV float byteToFloat(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyte<k>

// Four children at once.  n* / f* hold the quantised entry / exit planes of the four children on each axis (one byte
// each); a plane at byte value q lies at ray parameter q * s + b.  Then interval overlap with [tmin, tmax].
// fmaxf/fminf drop a NaN operand (0 * inf or inf - inf on an axis-parallel ray), which keeps the test conservative.
RT_DEV uint32_t testFour(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, V3 s, V3 b, float tmin, float tmax) {
    const f2_t sx = {s.x, s.x}, sy = {s.y, s.y}, sz = {s.z, s.z};
    const f2_t bx = {b.x, b.x}, by = {b.y, b.y}, bz = {b.z, b.z};
    uint32_t mask = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        f2_t a = {byteToFloat(nx, 2 * h), byteToFloat(nx, 2 * h + 1)}, bb = {byteToFloat(ny, 2 * h), byteToFloat(ny, 2 * h + 1)};
        f2_t c = {byteToFloat(nz, 2 * h), byteToFloat(nz, 2 * h + 1)};
        f2_t e = {byteToFloat(fx, 2 * h), byteToFloat(fx, 2 * h + 1)}, f = {byteToFloat(fy, 2 * h), byteToFloat(fy, 2 * h + 1)};
        f2_t g = {byteToFloat(fz, 2 * h), byteToFloat(fz, 2 * h + 1)};
        a = __builtin_elementwise_fma(a, sx, bx); bb = __builtin_elementwise_fma(bb, sy, by); c = __builtin_elementwise_fma(c, sz, bz);
        e = __builtin_elementwise_fma(e, sx, bx); f = __builtin_elementwise_fma(f, sy, by); g = __builtin_elementwise_fma(g, sz, bz);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float tn = fmaxf(fmaxf(a[k], bb[k]), fmaxf(c[k], tmin));
            float tf = fminf(fminf(e[k], f[k]), fminf(g[k], tmax));
            mask |= (tn <= tf) ? (1u << (2 * h + k)) : 0u;
        }
    }
    return mask;
}

// Conservative slab test of the eight child boxes of one node (head = the node's first 16 bytes); returns the bit mask
// of slots whose decoded box overlaps [tmin, tmax] along the ray (tmin is 0, or -inf for the mesh mode without a t > 0
// test).  Bits of unused slots are meaningless.  Conservative because the host pads every child box by more than the
// worst-case error of this form (rt_scene.cpp) before rounding it outwards onto the node's grid: a plane's parameter
// carries five roundings (subtraction, product, reciprocal 1 ulp, fma), each relative to the distance between the ray
// origin and the node - under 3e-7 of that distance on the plane's axis, and a hit that counts is nearer than 999.
RT_DEV uint32_t testChildren(u4_t head, u4_t qx, u4_t qy, u4_t qz, const SlabRay& r, float tmin, float tmax) {
    // plane at origin + q * step  ->  parameter (origin + q * step - o) / d = q * (step * rd) + (origin - o) * rd
    // (x and y in packed instructions: origin, reciprocal direction and the node's origin each lie in a register pair)
    const f2_t rdxy = {r.rd.x, r.rd.y}, oxy = {r.o.x, r.o.y};
    const f2_t stepxy = {ubits((head.w & 0xffu) << 23), ubits(((head.w >> 8) & 0xffu) << 23)}, nodexy = {ubits(head.x), ubits(head.y)};
    const f2_t sxy = stepxy * rdxy, bxy = (nodexy - oxy) * rdxy;
    const V3 s = mk(sxy.x, sxy.y, ubits(((head.w >> 16) & 0xffu) << 23) * r.rd.z);
    const V3 b = mk(bxy.x, bxy.y, (ubits(head.z) - r.o.z) * r.rd.z);
    // the ray enters through the low planes on axes it travels along in the positive direction, the high planes otherwise
    const bool px = !(r.oct & 1u), py = !(r.oct & 2u), pz = !(r.oct & 4u);
    uint32_t lo4 = testFour(px ? qx.x : qx.z, py ? qy.x : qy.z, pz ? qz.x : qz.z, px ? qx.z : qx.x, py ? qy.z : qy.x, pz ? qz.z : qz.x, s, b, tmin, tmax);
    uint32_t hi4 = testFour(px ? qx.y : qx.w, py ? qy.y : qy.w, pz ? qz.y : qz.w, px ? qx.w : qx.y, py ? qy.w : qy.y, pz ? qz.w : qz.y, s, b, tmin, tmax);
    return lo4 | (hi4 << 4);
}

// (x and y of a corner go through one packed subtraction and one packed multiplication: the corner arrives in a pair of scalar
// registers, origin and reciprocal direction lie in register pairs)
RT_DEV bool tightBoxHit(f4_t lo, f4_t hi, const SlabRay& r, float tmin, float tmax) {
    const f2_t oxy = {r.o.x, r.o.y}, rxy = {r.rd.x, r.rd.y};
    const f2_t loxy = {lo.x, lo.y}, hixy = {hi.x, hi.y};
    const f2_t a = (loxy - oxy) * rxy, b = (hixy - oxy) * rxy;
    float az = (lo.z - r.o.z) * r.rd.z, bz = (hi.z - r.o.z) * r.rd.z;
    float tn = fmaxf(fmaxf(fminf(a.x, b.x), fminf(a.y, b.y)), fmaxf(fminf(az, bz), tmin));
    float tf = fminf(fminf(fmaxf(a.x, b.x), fmaxf(a.y, b.y)), fminf(fmaxf(az, bz), tmax));
    return tn <= tf;
}

// Float -> unsigned with the same order (negative values below positive ones), and back.
RT_DEV uint32_t orderBits(float t) { const uint32_t b = fbits(t); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
RT_DEV float orderFloat(uint32_t m) { return ubits((m & 0x80000000u) ? (m & 0x7fffffffu) : ~m); }

// Traces the rays of the wave's lanes against every eligible model (kernels.cu:98-128 with Model::hit, Model.cpp:161-357).
//   ANYHIT = false: closest hit, merged into `best` (which must not hold a model hit yet) with the reference's tie rules:
//                   strictly nearer than any other kind of object found earlier; among triangles an exact tie goes to
//                   the lower model index, then the lower triangle index (the one the reference's loops meet first).
//   ANYHIT = true : returns true if some non-debug model has a triangle with 0 < t < 999.
// skipDebug: models flagged debug are not candidates (reflection and shadow rays).
// meshMode (kMesh*): which gates apply and whether a hit needs t > 0 - see Model.cpp:220-356.
// All 64 lanes must call this together; hasRay says whether the lane brings a ray.
//
// The work of all rays of the wave sits in two shared pools in LDS: pending children of visited nodes, and pending runs
// of triangles.  Every turn the wave takes the 64 newest entries of one pool, one per lane, whichever rays they belong
// to; a lane fetches the ray of its entry from LDS.  (With one ray per lane and a private stack, 35 of the 60 lanes that
// entered the traversal were still busy in an average turn, 27 in the slowest waves: rays of one 8x8 block need very
// different numbers of steps.  Here a ray with much work gets the lanes of the rays that are done; 47 entries are taken
// in an average turn.)
//
// The nearest hit of ray l is kept in bestKey[l] = orderBits(t) << 32 | key, lowered with a 64-bit LDS atomic minimum;
// key = the triangle's index in the scene-wide attribute array, which grows with (model, triangle id), so that the
// minimum implements the tie rule.  The start value holds the distance of the nearest other object with key 0: a triangle
// must be strictly nearer.  bestRec[l] remembers the triangle record of the minimum; when the pools are empty each lane
// intersects its winning triangle once more to obtain t, u and v (same arithmetic, same values).  An any-hit ray that
// found something stores 0, and its remaining entries are dropped when they come up.
//
// Entries of one ray can be taken in the same turn, so subtrees are entered that a strictly front-to-back traversal
// would have pruned (6 % more box tests, 10 % more triangle tests on the 1080p benchmark scene); what is visited beyond
// need never changes the result.  Newest-first order keeps that small: the top of the pool are the children just found,
// nearest child first.
// Pool capacity: a lane's node turn takes one entry and adds at most two (remaining siblings, new children), so a full
// turn needs 64 free entries; with fewer than 88 free, only as many lanes take entries as leaves 24 in reserve, and at
// the limit single lanes take turns (depth-first on the newest entry, which needs at most one more entry per tree
// level) until there is room again.  A node turn adds at most one entry per lane to the pool of triangle work (the set
// of leaves the ray hit in the node) and is only taken while that pool has room for 64 and holds fewer than RT_TRI_TURN_AT
// entries; the triangle turns take the leaves off a set one by one, nearest first.
// TIMED: the call adds to the wave's section clocks and turn counts (instrumented kernel; off for its counting-only second pass).
template <bool ANYHIT, bool STATS, bool TIMED = STATS>
RT_DEV bool tracePool(SceneK& S, bool hasRay, V3 o, V3 d, bool cull, bool skipDebug, int meshMode, uint32_t poolCapacity, uint32_t trav, Closest& best, WaveLds lds, Counters& cnt) {
    const uint32_t forestRoot = S.forestRoot;
    if (forestRoot == kNoRoot) return false;
    const uint32_t nodeCapacity = poolCapacity & 0xffffu, triCapacity = poolCapacity >> 16;
    SectionTimer<TIMED> timerWhole;
    const uint32_t lane = threadIdx.x & 63u;
    const int nMdl = S.nMdl;
    const bool positiveOnly = meshMode != kMeshNoBox;
    const float tmin = positiveOnly ? 0.0f : -__builtin_inff();
    const bool pregated = nMdl <= kPregatedModels;

    // Is model mm a candidate for the ray at all: not hidden from this ray type, and (whole-box mode) through the
    // reference's exact bounding-box gate.
    // (uniform: mm is the same for the whole wave)
    // (implied: the ray met the model's tight box and starts within the range in which that settles the gate, see the w word
    // of DevScene::mdlTightMin)
    auto gate = [&](const F4* mdlMat, const F4* gateMin, const F4* gateMax, int mm, V3 ro, V3 rdir, V3 rrd, bool fin, const bool uniform, bool implied) -> bool {
        if ((skipDebug || ANYHIT) && ((uniform ? lduFlags(mdlMat + 3 * mm + 2) : fbits(mdlMat[3 * mm + 2].w)) & 1u)) return false;
        if (meshMode == kMeshWholeBox) {
            if (implied && fin) return true;
            if (STATS) cnt.gateTests++;
            const f4_t lo = uniform ? ldu4(gateMin + mm) : ld4(gateMin + mm), hi = uniform ? ldu4(gateMax + mm) : ld4(gateMax + mm);
            return boxHitDecision(xyz(lo), xyz(hi), ro, rdir, rrd, fin);
        }
        return true;
    };

    const float startT = ANYHIT ? FAR_T : best.t;
    const unsigned long long init = (unsigned long long)orderBits(startT) << 32;
    // Small scenes: every lane evaluates the gates of its own ray up front, all lanes with the same trip count (the exact
    // gate would otherwise be met by single lanes at model roots).  The cheap conservative test of the triangles' own box
    // comes first; the order does not change the result.  Larger scenes evaluate a gate when an entry reaches a model root.
    unsigned long long gateBits = 0;
    const SlabRay own = makeSlabRay(o, d);
    if (hasRay && pregated) {
        SectionTimer<TIMED> timerG;
        const SlabRay& sr = own;
        const bool fin = fmaxf(fmaxf(fabsf(sr.rd.x), fabsf(sr.rd.y)), fabsf(sr.rd.z)) < __builtin_inff();
        const F4* const tightMin = inGlobal(S.mdlTightMin), * const tightMax = inGlobal(S.mdlTightMax);
        const F4* const mdlMat = inGlobal(S.mdlMat), * const gateMin = inGlobal(S.mdlGateMin), * const gateMax = inGlobal(S.mdlGateMax);
        // (quieted once here: the minimum / maximum instructions would otherwise do it for these two operands in every round)
        const float tminGate = __builtin_canonicalizef(tmin), tmaxGate = __builtin_canonicalizef(startT);
        const float originReach = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
        for (int j = 0; j < nMdl; j++) {
            const f4_t tightLo = ldu4(tightMin + j);
            if (!tightBoxHit(tightLo, ldu4(tightMax + j), sr, tminGate, tmaxGate)) continue;
            if (gate(mdlMat, gateMin, gateMax, j, o, d, sr.rd, fin, true, originReach <= tightLo.w)) gateBits |= 1ull << j;
        }
        timerG.stop(lds, 2);
    }
    const bool enters = hasRay && (!pregated || gateBits != 0);
    if (__ballot(enters) == 0ull) { timerWhole.stop(lds, 3); return false; }     // no ray of this wave comes near a mesh
    {
        f4_t ro4 = {o.x, o.y, o.z, ubits((uint32_t)gateBits)}, rd4 = {d.x, d.y, d.z, ubits((uint32_t)(gateBits >> 32))};
        f4_t rr4 = {own.rd.x, own.rd.y, own.rd.z, ubits(own.oct)};
        lds.rayO[lane] = ro4; lds.rayD[lane] = rd4;
        if (RT_RCP_IN_LDS) lds.rayR[lane] = rr4;
        lds.bestKey[lane] = init;
    }
    // what the turns below read through, for the whole loop in scalar registers
    const F4* const nodes = inGlobal(S.nodes), * const tris = inGlobal(S.tris);
    const uint32_t* const mdlInfo = inGlobal(S.mdlInfo);
    // The octant in an entry orders the hit children and leaves front to back.  A ray that only asks whether anything is hit
    // gains nothing from the order: its entries carry octant 0 (slot order) and the masks of a node are not permuted for it;
    // which planes of a child box the ray enters through comes from the ray's own octant, kept beside its reciprocal direction.
    const uint32_t entryOct = (ANYHIT && RT_RCP_IN_LDS) ? 0u : own.oct;
    // one pseudo entry per ray that resolves to the forest root
    uint32_t nodeCount, triCount = 0;
    if ((trav & kTravDirectStart) && pregated && nMdl <= kDirectStartModels && (uint32_t)nMdl * 64u + 88u <= nodeCapacity) {
        // few models, gates known: one entry per ray and model whose gate the ray passed, resolving to that model's root
        nodeCount = 0;
        for (int j = 0; j < nMdl; j++) {
            const uint32_t root = lduw(mdlInfo + 4 * j);
            const bool in = enters && ((gateBits >> j) & 1ull) != 0 && root != kNoRoot;
            const unsigned long long m = __ballot(in);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (in) lds.nodePool[nodeCount + rank] = (unsigned long long)(root | (entryOct << kNodeIndexBits)) | ((unsigned long long)((1u << 8) | (lane << 16) | (0xffu << 24)) << 32);
            nodeCount += (uint32_t)__builtin_popcountll(m);
        }
    } else {
        const unsigned long long m = __ballot(enters);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (enters) lds.nodePool[rank] = (unsigned long long)(forestRoot | (entryOct << kNodeIndexBits)) | ((unsigned long long)((1u << 8) | (lane << 16) | (0xffu << 24)) << 32);
        nodeCount = (uint32_t)__builtin_popcountll(m);
    }
    __builtin_amdgcn_wave_barrier();

    while (nodeCount != 0 || triCount != 0) {
        // (the two counts are the same in every lane; said explicitly, because otherwise the arithmetic on them ends up in
        // vector instructions)
        nodeCount = (uint32_t)__builtin_amdgcn_readfirstlane((int)nodeCount); triCount = (uint32_t)__builtin_amdgcn_readfirstlane((int)triCount);
        // a node turn adds at most one triangle entry per lane
        const bool nodeTurn = nodeCount != 0 && triCount < RT_TRI_TURN_AT && triCount + 65u <= triCapacity;
        if (nodeTurn) {
            SectionTimer<TIMED> timerA;
            fetchPriority(true);
            const uint32_t room = nodeCapacity - nodeCount;
            // a lane's turn adds at most one entry net; 24 entries stay in reserve for a depth-first descent by single lanes
            const uint32_t width = room >= 88u ? 64u : (room > 25u ? room - 24u : 1u);
            const uint32_t take = nodeCount < width ? nodeCount : width;
            const bool mine = lane < take;
            if (TIMED && lane == 0) { lds.prof[14] += take; lds.prof[15] += 1ull + (64ull << 32); }
            unsigned long long e = 0;
            if (mine) e = lds.nodePool[nodeCount - 1u - lane];
            nodeCount -= take;
            // the low word names the children and the ray's octant, so the record of the nearest hit child can be requested
            // before the ray itself has arrived from LDS
            const uint32_t low = (uint32_t)e, packed = (uint32_t)(e >> 32);
            const uint32_t base = low & kNodeIndexMask, oct = low >> kNodeIndexBits;
            const uint32_t imask = packed & 0xffu, hits = (packed >> 8) & 0xffu, owner = (packed >> 16) & 63u;
            uint32_t chunkMask = packed >> 24;
            const uint32_t visited = base + __builtin_popcount(imask & ((1u << ((uint32_t)__builtin_ctz(hits | 0x100u) ^ oct)) - 1u));
            const F4* np = nodes + (size_t)visited * kNodeFloat4s;
            u4_t head, meta, qx, qy, qz;      // read by the lanes that loaded them
            loadsHaveArrived();               // (the previous turn's: nothing older is outstanding when this turn's loads go out)
            if (mine) {
                head = *reinterpret_cast<const u4_t*>(np);
                // (three words: the fourth is not read, and a register that waits for it would be handed to the next address
                // computation, which would then wait for the load)
                { typedef uint32_t u3_t __attribute__((ext_vector_type(3))); const u3_t m3 = *reinterpret_cast<const u3_t*>(np + 1); meta = u4_t{m3.x, m3.y, m3.z, 0u}; }
                qx = *reinterpret_cast<const u4_t*>(np + 2); qy = *reinterpret_cast<const u4_t*>(np + 3); qz = *reinterpret_cast<const u4_t*>(np + 4);
            }
            fetchPriority(false);
            unsigned long long bk = 0;
            if (mine) bk = lds.bestKey[owner];
            const bool alive = mine && !(ANYHIT && bk == 0ull);
            // the siblings that are still waiting go back first, so that the children found now end up on top
            const uint32_t rest = hits & (hits - 1u);
            {
                const bool again = alive && rest != 0u;
                const unsigned long long m = __ballot(again);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (again) lds.nodePool[nodeCount + rank] = (unsigned long long)low | ((unsigned long long)(imask | (rest << 8) | (owner << 16) | (chunkMask << 24)) << 32);
                nodeCount += (uint32_t)__builtin_popcountll(m);
            }
            // childBase and childImask are read only where childHits is not zero - where a lane has set them below (no start
            // values: they would be written by every lane every turn)
            uint32_t childHits = 0, pend = 0, childBase, childImask;
            if (alive) {
                const f4_t ro4 = lds.rayO[owner];
                const V3 ro = xyz(ro4);
                SlabRay sr;
                f4_t rdHere = {0, 0, 0, 0};
                if (RT_RCP_IN_LDS) {
                    const f4_t rr4 = lds.rayR[owner];
                    sr.rd = xyz(rr4); sr.o = ro; sr.oct = ANYHIT ? fbits(rr4.w) : oct;
                } else {
                    rdHere = lds.rayD[owner];
                    sr = makeSlabRay(ro, xyz(rdHere));
                }
                const float tmax = ANYHIT ? FAR_T : orderFloat((uint32_t)(bk >> 32));
                uint32_t hmask = testChildren(head, qx, qy, qz, sr, tmin, tmax);
                if (meta.z != kNoModel) {       // root of a model's octree: the model's gates decide whether the ray enters it
                    const int mm = (int)meta.z;
                    const f4_t rd4 = RT_RCP_IN_LDS ? lds.rayD[owner] : rdHere;
                    const V3 rdir = xyz(rd4);
                    const bool fin = fmaxf(fmaxf(fabsf(sr.rd.x), fabsf(sr.rd.y)), fabsf(sr.rd.z)) < __builtin_inff();
                    const unsigned long long bits = (unsigned long long)fbits(ro4.w) | ((unsigned long long)fbits(rd4.w) << 32);
                    bool enter = pregated ? (((bits >> mm) & 1ull) != 0) : gate(inGlobal(S.mdlMat), inGlobal(S.mdlGateMin), inGlobal(S.mdlGateMax), mm, ro, rdir, sr.rd, fin, false, false);
                    if (enter && meshMode == kMeshEightBoxes) {      // Model.cpp:225-227: each chunk of triangles has its own exact gate
                        chunkMask = 0;
                        const F4* const chunkMin = inGlobal(S.mdlChunkMin), * const chunkMax = inGlobal(S.mdlChunkMax);
                        for (int b = 0; b < 8; b++) {
                            if (STATS) cnt.gateTests++;
                            if (boxHitDecision(xyz(ld4(chunkMin + 8 * mm + b)), xyz(ld4(chunkMax + 8 * mm + b)), ro, rdir, sr.rd, fin)) chunkMask |= 1u << b;
                        }
                        enter = chunkMask != 0;
                    }
                    if (!enter) hmask = 0;
                }
                const uint32_t nodeImask = meta.y & 0xffu, nodeLmask = (meta.y >> 8) & 0xffu;
                if (STATS) { cnt.nodeTests += __builtin_popcount(nodeImask | nodeLmask); cnt.nodeVisits++; }
                childBase = meta.x | (oct << kNodeIndexBits); childImask = nodeImask;
                // meta.y holds the internal mask in byte 0 and the leaf mask in byte 1: both are permuted at once
                const uint32_t bothSlots = (hmask | (hmask << 8)) & meta.y & 0xffffu;
                const uint32_t both = (ANYHIT && RT_RCP_IN_LDS) ? bothSlots : permuteOctants(bothSlots, oct);
                childHits = both & 0xffu;
                pend = both >> 8;
            }
            // While the pool holds fewer entries than the wave has lanes, every hit child gets an entry of its own (farthest
            // first, so that the nearest ends up on top) and the next turn visits siblings side by side instead of one per turn.
            bool expand = false;
            // (the limit is checked first: a pool that is already larger cannot qualify, and the eight ballots are skipped)
            const uint32_t expandLimit = min(((trav >> 8) & 0xffu) * 8u, nodeCapacity - 24u);
            if ((trav & kTravExpandSiblings) && nodeCount <= expandLimit) {
                const uint32_t total = nodeCount + waveSum((uint32_t)__builtin_popcount(childHits));
                // the entries written below end before position `total`: never beyond the pool, whatever limit the caller set
                expand = total <= expandLimit;
            }
            if (expand) {
                uint32_t waiting = childHits;
                for (;;) {
                    const bool more = waiting != 0u;
                    const unsigned long long m = __ballot(more);
                    if (m == 0ull) break;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (more) {
                        const uint32_t k = 31u - (uint32_t)__builtin_clz(waiting);
                        waiting &= ~(1u << k);
                        lds.nodePool[nodeCount + rank] = (unsigned long long)childBase | ((unsigned long long)(childImask | (1u << (k + 8u)) | (owner << 16) | (chunkMask << 24)) << 32);
                    }
                    nodeCount += (uint32_t)__builtin_popcountll(m);
                }
            } else {
                const bool down = childHits != 0u;
                const unsigned long long m = __ballot(down);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (down) lds.nodePool[nodeCount + rank] = (unsigned long long)childBase | ((unsigned long long)(childImask | (childHits << 8) | (owner << 16) | (chunkMask << 24)) << 32);
                nodeCount += (uint32_t)__builtin_popcountll(m);
            }
            // the leaves the ray hit in this node: one entry that names the node and the set (lowest bit = nearest leaf); the
            // triangle turns take the leaves off it one by one and look their triangles up in the node record
            {
                const bool has = pend != 0u;
                const unsigned long long m = __ballot(has);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (has) lds.triPool[triCount + rank] = (unsigned long long)visited | ((unsigned long long)(owner | (chunkMask << 8) | (pend << 16) | (oct << 24) | kLeafSet) << 32);
                triCount += (uint32_t)__builtin_popcountll(m);
            }
            loadsHaveArrived();
            timerA.stop(lds, 0);
        } else {
            SectionTimer<TIMED> timerB;
            fetchPriority(true);
            // A lane takes one entry and puts back at most two (the rest of a leaf set, the rest of a run of triangles): a full
            // turn needs 64 free entries; with fewer, one entry stays free, and in a full pool the newest entry is a run, which
            // a single lane takes without needing room (see the order of the two pushes below).
            const uint32_t room = triCapacity - triCount;
            const uint32_t width = room >= 64u ? 64u : (room >= 2u ? room - 1u : 1u);
            const uint32_t take = triCount < width ? triCount : width;
            const bool mine = lane < take;
            if (TIMED && lane == 0) { lds.prof[14] += take; lds.prof[15] += 1ull + (64ull << 32); }
            unsigned long long e = 0;
            if (mine) e = lds.triPool[triCount - 1u - lane];
            triCount -= take;
            const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32);
            const uint32_t owner = hi & 63u, chunks = (hi >> 8) & 0xffu;
            unsigned long long bk = 0;
            if (mine) bk = lds.bestKey[owner];
            const bool alive = mine && !(ANYHIT && bk == 0ull);
            // a leaf set: the nearest leaf is taken now (its first triangle record stands in the node), the others go back
            uint32_t first = lo, restMask = 0;
            if (alive && (hi & kLeafSet)) {
                const uint32_t mask = (hi >> 16) & 0xffu;
                const uint32_t lc = (uint32_t)__builtin_ctz(mask) ^ ((hi >> 24) & 7u);
                first = reinterpret_cast<const uint32_t*>(nodes + (size_t)lo * kNodeFloat4s + 5)[lc];
                restMask = mask & (mask - 1u);
            }
            {
                const bool again = restMask != 0u;
                const unsigned long long m = __ballot(again);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (again) lds.triPool[triCount + rank] = (unsigned long long)lo | ((unsigned long long)((hi & ~(0xffu << 16)) | (restMask << 16)) << 32);
                triCount += (uint32_t)__builtin_popcountll(m);
            }
            bool goesOn = false;
            if (alive) {
                const V3 ro = xyz(lds.rayO[owner]), rdir = xyz(lds.rayD[owner]);
                const F4* tp = tris + (size_t)first * 3;
                loadsHaveArrived();           // (the look-up of `first`, and whatever a previous turn left unread)
                // (a record's second float4 ends with the model index, which only the final hit needs)
                const f4_t A0 = ld4(tp), C0 = ld4(tp + 2), A1 = ld4(tp + 3), C1 = ld4(tp + 5);
                const f3_t B0 = ld3(tp + 1), B1 = ld3(tp + 4);
                fetchPriority(false);
                auto testTriangle = [&](f4_t A, f3_t B, f4_t C, uint32_t record) {
                    const uint32_t idw = fbits(A.w);
                    if (STATS) cnt.triTests++;
                    // Moeller-Trumbore exactly as the reference evaluates it (edges were subtracted on the host)
                    V3 v0 = xyz(A), e1 = mk(B.x, B.y, B.z), e2 = xyz(C);
                    V3 pvec = cross(rdir, e2);
                    float det = dot(e1, pvec);
                    bool rejected = (cull && det >= -0.01f) || (fabsf(det) <= 0);
                    if (meshMode == kMeshEightBoxes) rejected = rejected || (idw & kTriChunkDropped) || !((chunks >> ((idw >> kTriChunkShift) & 7u)) & 1u);
                    if (rejected) return;
                    float invDet = 1 / det;
                    V3 tvec = ro - v0;
                    float u = dot(tvec, pvec) * invDet;
                    if (u < 0 || u > 1) return;
                    V3 qvec = cross(tvec, e1);
                    float v = dot(rdir, qvec) * invDet;
                    if (v < 0 || u + v > 1) return;
                    float t = dot(e2, qvec) * invDet;
                    const bool inRange = !positiveOnly || t > 0;      // Model.cpp:287 vs :327
                    if (!inRange || !(t == t)) return;
                    if (ANYHIT) {
                        if (t < FAR_T) lds.bestKey[owner] = 0ull;
                        return;
                    }
                    // (the third record carries the model's first position in the attribute array)
                    const uint32_t key = fbits(C.w) + (idw & kTriIdMask);
                    const unsigned long long cand = ((unsigned long long)orderBits(t + 0.0f) << 32) | key;    // -0 orders as +0
                    if (cand < bk) {
                        atomicMin(&lds.bestKey[owner], cand);
                        // every lane of this turn has applied its minimum by now: only the lane that holds it sees its own value
                        if (__hip_atomic_load(&lds.bestKey[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == cand) lds.bestRec[owner] = record;
                    }
                };
                testTriangle(A0, B0, C0, first);
                uint32_t idw = fbits(A0.w);
                if (!(idw & kTriLast)) { testTriangle(A1, B1, C1, first + 1u); idw = fbits(A1.w); }
                goesOn = !(idw & kTriLast);
            }
            {
                const unsigned long long m = __ballot(goesOn);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (goesOn) lds.triPool[triCount + rank] = (unsigned long long)(first + 2u) | ((unsigned long long)(owner | (chunks << 8)) << 32);
                triCount += (uint32_t)__builtin_popcountll(m);
            }
            fetchPriority(false);
            loadsHaveArrived();
            timerB.stop(lds, 1);
        }
        __builtin_amdgcn_wave_barrier();
    }

    bool found = false;
    if (hasRay) {
        const unsigned long long bk = lds.bestKey[lane];
        if (ANYHIT) {
            found = bk == 0ull;
        } else if (bk != init) {
            // the winning triangle once more, for t, u and v
            // (the lane number passes through an empty asm statement so that this address is formed here: formed once at the top of
            // the kernel, it would occupy a register through every trace of the pixel)
            uint32_t self = lane; asm volatile("" : "+v"(self));
            const uint32_t record = lds.bestRec[self];
            const F4* tp = tris + (size_t)record * 3;
            const f4_t A = ld4(tp), B = ld4(tp + 1), C = ld4(tp + 2);
            V3 v0 = xyz(A), e1 = xyz(B), e2 = xyz(C);
            V3 pvec = cross(d, e2);
            float det = dot(e1, pvec);
            float invDet = 1 / det;
            V3 tvec = o - v0;
            float u = dot(tvec, pvec) * invDet;
            V3 qvec = cross(tvec, e1);
            float v = dot(d, qvec) * invDet;
            float t = dot(e2, qvec) * invDet;
            best.t = t; best.kind = KIND_MODEL; best.idx = (int)fbits(B.w); best.tri = (int)(fbits(C.w) + (fbits(A.w) & kTriIdMask)); best.u = u; best.v = v;
        }
    }
    __builtin_amdgcn_wave_barrier();
    timerWhole.stop(lds, 3);
    return found;
}

// ---------------------------------------------------------------- linear object traces

// kernels.cu:67-96 over the three analytic object tables, closest hit.  skipDebug* mirror the ray types the
// reference uses for each table at this level.
// flops (instrumented kernel only): 30 / 14 / 26 per sphere / plane / box test that is made.
template <bool STATS>
RT_DEV void traceAnalytic(SceneK& S, V3 o, V3 d, bool skipDebugSpheres, bool tracePlanes, bool skipDebugPlanes,
                          bool traceBoxes, Closest& best, uint32_t& flops) {
    const int nSph = S.nSph;
    const F4* const sphGeom = inGlobal(S.sphGeom), * const sphMat = inGlobal(S.sphMat);
    for (int i = 0; i < nSph; i++) {
        f4_t g = ldu4(sphGeom + i);
        uint32_t flags = lduFlags(sphMat + 3 * i + 2);
        if (flags & 2u) continue;
        // (the reference tests the object and then discards a debug object's hit, kernels.cu:75-83; the test has no other
        // effect, and the flags are the same for the whole wave, so the discarded tests are not made at all - the light's
        // marker sphere of every reference scene is such an object for every shadow and reflection ray)
        if (skipDebugSpheres && (flags & 1u)) continue;
        if (STATS) flops += 30u;
        float t0;
        if (sphereHit(xyz(g), g.w, o, d, t0)) {
            if (t0 < best.t) { best.t = t0; best.kind = KIND_SPHERE; best.idx = i; }
        }
    }
    if (tracePlanes) {
        const int nPln = S.nPln;
        const F4* const plnP = inGlobal(S.plnP), * const plnN = inGlobal(S.plnN), * const plnMat = inGlobal(S.plnMat);
        for (int i = 0; i < nPln; i++) {
            uint32_t flags = lduFlags(plnMat + 3 * i + 2);
            if (flags & 2u) continue;
            if (skipDebugPlanes && (flags & 1u)) continue;
            if (STATS) flops += 14u;
            float t0;
            if (planeHit(xyz(ldu4(plnP + i)), xyz(ldu4(plnN + i)), o, d, t0)) {
                if (t0 < best.t) { best.t = t0; best.kind = KIND_PLANE; best.idx = i; }
            }
        }
    }
    if (traceBoxes) {
        const int nBox = S.nBox;
        const F4* const boxMin = inGlobal(S.boxMin), * const boxMax = inGlobal(S.boxMax), * const boxMat = inGlobal(S.boxMat);
        for (int i = 0; i < nBox; i++) {
            if (lduFlags(boxMat + 3 * i + 2) & 1u) continue;   // boxes are always traced as reflection rays (kernels.cu:228)
            if (STATS) flops += 26u;
            float t0;
            if (boxHit(xyz(ldu4(boxMin + i)), xyz(ldu4(boxMax + i)), o, d, t0)) {
                if (t0 < best.t) { best.t = t0; best.kind = KIND_BOX; best.idx = i; }
            }
        }
    }
}

template <bool STATS>
RT_DEV bool shadowSpheres(SceneK& S, V3 o, V3 d, uint32_t& flops) {
    const int nSph = S.nSph;
    const F4* const sphGeom = inGlobal(S.sphGeom), * const sphMat = inGlobal(S.sphMat);
    for (int i = 0; i < nSph; i++) {
        f4_t g = ldu4(sphGeom + i);
        uint32_t flags = lduFlags(sphMat + 3 * i + 2);
        if (flags & 3u) continue;       // hidden, or a debug object, which never casts a shadow (kernels.cu:78)
        if (STATS) flops += 30u;
        float t0;
        if (sphereHit(xyz(g), g.w, o, d, t0)) return true;
    }
    return false;
}

template <bool STATS>
RT_DEV bool shadowBoxes(SceneK& S, V3 o, V3 d, uint32_t& flops) {
    const int nBox = S.nBox;
    if (nBox == 0) return false;
    const F4* const boxMin = inGlobal(S.boxMin), * const boxMax = inGlobal(S.boxMax), * const boxMat = inGlobal(S.boxMat);
    for (int i = 0; i < nBox; i++) {
        if (lduFlags(boxMat + 3 * i + 2) & 1u) continue;
        if (STATS) flops += 26u;
        float t0;
        if (boxHit(xyz(ldu4(boxMin + i)), xyz(ldu4(boxMax + i)), o, d, t0)) return true;
    }
    return false;
}

// ---------------------------------------------------------------- shading

// v = normalise(cameraPosition) (kernels.cu:20-65 evaluates it per pixel; it is the same for the whole frame and arrives in
// the frame parameters, computed on the host with the same three divisions and square root).
// L = normalise(lightPos - point), which the caller also needs as the direction of the hard shadow ray (kernels.cu:22 and :269
// evaluate the same expression).
// Where a wave's pixels lie: the frame coordinates of its lane 0 (wave-uniform, kept in scalar registers for the whole kernel).
// A lane's own pixel is worked out from them where it is needed - for the camera ray at the start, for the sample seeds of
// soft shadows, for the stores at the end - instead of being kept in two vector registers from the first use to the last
// (which, at 128 registers, were spilled: every wave wrote its scratch, and the lines went back to HBM after it).
struct PixelBase { uint32_t x0, y0; bool rows; };      // rows: lane l is pixel (x0 + (l & 7), y0 + (l >> 3)); else (x0 + (l & 7), y0)
RT_DEV int pixelX(PixelBase b) {
    uint32_t x0 = b.x0; asm volatile("" : "+s"(x0));       // opaque: worked out afresh at every use
    return (int)(x0 + (threadIdx.x & 7u));
}
RT_DEV int pixelY(PixelBase b) {
    uint32_t y0 = b.y0; asm volatile("" : "+s"(y0));
    return (int)(y0 + (b.rows ? ((threadIdx.x & 63u) >> 3) : 0u));
}

RT_DEV V3 lighting(const Mat& mat, V3 lightPos, V3 lightCol, V3 point, V3 L, V3 v, V3 normal, FrameK& P, PixelBase pixel) {
    V3 colour = mk(0, 0, 0);
    const uint32_t flags = P.flags;
    if (flags & kCfgAmbient) colour = (lightCol * mat.ka) * mat.ambient;
    if (flags & kCfgDiffuse) {
        float NdotL = selMax(dot(normal, L), 0.0f);
        colour = colour + ((NdotL > 0.0f) ? (((mat.kd * lightCol) * mat.diffuse) * NdotL) : mk(0, 0, 0));
    }
    // The specular term is (lightCol * specular) * pow(max(r.v, 0), shininess).  With specular = 0, a finite light colour and a
    // shininess in [0, 1e6] (the cosine is at most 1 plus rounding, so the power is finite) it is a zero of either sign, and adding
    // it leaves every pixel as it is: the floor, the light marker and the diffuse models of the reference's scenes have such
    // materials.  The reflection vector and the power are skipped when no lane of the wave that is being shaded needs them.
    const bool lightFinite = fabsf(lightCol.x) < __builtin_inff() && fabsf(lightCol.y) < __builtin_inff() && fabsf(lightCol.z) < __builtin_inff();
    const bool specularCounts = !(mat.specular == 0.0f && mat.shininess >= 0.0f && mat.shininess <= 1.0e6f) || !lightFinite;
    if ((flags & kCfgSpecular) && __ballot(specularCounts) != 0ull) {
        if (flags & kCfgAreaSpecular) {
            const int softNum = P.softNum;
            const float softRadius = P.softRadius;
            const int px = pixelX(pixel), py = pixelY(pixel);
            for (int i = 0; i < softNum; i++) {
                V3 ps = lightSampleOffset(sampleSeed(px, py, i), softRadius);
                V3 Ls = normalise((lightPos + ps) - point);
                V3 r = normalise(neg(reflect(Ls, normal)));
                float RdotV = selMax(dot(r, v), 0.0f);
                colour = colour + (lightCol * mat.specular) * powSpecular(RdotV, mat.shininess);
            }
        } else {
            V3 r = normalise(neg(reflect(L, normal)));
            float RdotV = selMax(dot(r, v), 0.0f);
            colour = colour + (lightCol * mat.specular) * powSpecular(RdotV, mat.shininess);
        }
    }
    return colour;
}

// Everything the shader needs about a closest hit.
struct Surface {
    Mat mat;
    V3 point, normal;   // normal as the reference holds it after its trace calls (see normaliseCount)
};

// Rebuilds hit point, normal and material of `h`.  `renorm` = how many trace routines of the reference run from
// the winning object's own routine to the last one: each of them normalises the stored normal once more
// (kernels.cu:90-93, 122-125).
// A plane's normal is the same for all of its pixels: it is looked up (DevScene::plnNormals), and with it `lightNormal`, the
// normal once more normalised, as the lighting takes it (kernels.cu:240-245); for every other kind of object the caller
// normalises per pixel.
template <bool STATS>
RT_DEV Surface resolveSurface(SceneK& S, V3 o, V3 d, const Closest& h, int renorm, Counters& cnt, V3& lightNormal) {
    Surface s;
    s.point = o + d * h.t;
    if (h.kind == KIND_PLANE) {
        s.mat = loadMat(inGlobal(S.plnMat), h.idx);
        const F4* const normals = inGlobal(S.plnNormals) + 5 * h.idx + renorm;
        const f3_t n0 = ld3(normals), n1 = ld3(normals + 1);      // (three words each: see ld3)
        s.normal = mk(n0.x, n0.y, n0.z);
        lightNormal = mk(n1.x, n1.y, n1.z);
        return s;
    }
    if (h.kind == KIND_SPHERE) {
        s.mat = loadMat(inGlobal(S.sphMat), h.idx);
        s.normal = s.point - xyz(ld4(inGlobal(S.sphGeom) + h.idx));
    } else if (h.kind == KIND_BOX) {
        s.mat = loadMat(inGlobal(S.boxMat), h.idx);
        s.normal = boxNormalAt(xyz(ld4(inGlobal(S.boxMin) + h.idx)), xyz(ld4(inGlobal(S.boxMax) + h.idx)), s.point);
    } else {
        s.mat = loadMatWithoutColours(inGlobal(S.mdlMat), h.idx);
        const F4* ap = inGlobal(S.attrs) + (size_t)(uint32_t)h.tri * kAttrFloat4s;
        f4_t a0 = ld4(ap), a1 = ld4(ap + 1), a2 = ld4(ap + 2), a3 = ld4(ap + 3), a4 = ld4(ap + 4), a5 = ld4(ap + 5), a6 = ld4(ap + 6);
        if (STATS) cnt.attrFetches++;
        float u = h.u, v = h.v, w = 1 - u - v;
        // Model.cpp:300-310: u weights the first vertex, v the second, 1-u-v the third
        s.normal = (xyz(a0) * u + xyz(a1) * v) + xyz(a2) * w;
        V3 ka0 = mk(a0.w, a1.w, a2.w), ka1 = xyz(a3), ka2 = xyz(a4);
        V3 kd0 = mk(a3.w, a4.w, a5.w), kd1 = xyz(a5), kd2 = xyz(a6);
        s.mat.ka = (ka0 * u + ka1 * v) + ka2 * w;
        s.mat.kd = (kd0 * u + kd1 * v) + kd2 * w;
    }
    for (int i = 0; i < renorm; i++) s.normal = normalise(s.normal);
    return s;
}

RT_DEV int renormCount(int kind, bool planesRun, bool boxesRun, bool modelsRun) {
    if (kind == KIND_MODEL) return 1;
    const int afterBoxes = modelsRun ? 1 : 0;
    if (kind == KIND_BOX) return 1 + afterBoxes;
    const int afterPlanes = (boxesRun ? 1 : 0) + afterBoxes;
    if (kind == KIND_PLANE) return 1 + afterPlanes;
    return 1 + (planesRun ? 1 : 0) + afterPlanes;   // sphere
}

// One camera sample: kernels.cu:224-318 with the reflection chain kernels.cu:130-222 unrolled into four levels (primary
// ray, then up to three reflections).  Every mesh trace sits in wave-uniform control flow: all 64 lanes call tracePool,
// and a lane without a ray at that point - outside the frame, chain ended, nothing to shade - passes hasRay = false and
// lends its lane to the others.  `live`: the lane has a camera sample at all.
template <bool STATS, bool IDS>
RT_DEV V3 castSample(SceneK& S, FrameK& P, CamK cam, bool live, V3 o, V3 d, PixelBase pixel, WaveLds lds, Counters& cnt, Closest& primaryOut) {
    const V3 background = mk(P.background);
    V3 lvl0 = background, lvl1 = background, lvl2 = background, lvl3 = background;
    float str0 = 0, str1 = 0, str2 = 0;
    // per-lane flags of the levels in one register (as booleans each would be a lane mask in a pair of scalar registers, kept
    // across every trace of the sample): bit l = level l is shadowed, bit 4 + l = level l + 1 is added to level l, bit 8 = soft
    uint32_t levelBits = 0;
    float softFactor = 1.0f;
    int deepest = 0;

    bool active = live;
    V3 ro = o, rdir = d;
#pragma unroll 1
    for (int level = 0; level < 4; level++) {
        if (__ballot(active) == 0ull) break;
        // the settings are read once per level, so that nothing of them has to be kept across levels
        const uint32_t flags = P.flags;
        const bool reflections = (flags & kCfgReflections) != 0, reflectPlanes = (flags & kCfgReflectPlanes) != 0;
        const bool boxesRun = (flags & kCfgRenderBoxes) != 0, modelsRun = (flags & kCfgRenderModels) != 0;
        const bool modelsVisible = modelsRun && (flags & kCfgModelsVisible) != 0;
        const bool hardShadows = (flags & kCfgHardShadows) != 0, cull = (flags & kCfgCull) != 0;
        // capacities of the two pools: the compiled ones unless the frame parameters ask for less (tests of the low-room paths)
        const uint32_t askPools = (uint32_t)P.nodePoolEntries, askNodes = askPools & 0xffffu, askTris = askPools >> 16;
        const uint32_t nodeCapacity = ((askNodes >= 96u && askNodes < (uint32_t)kNodePool) ? askNodes : (uint32_t)kNodePool)
                                    | (((askTris >= 72u && askTris < (uint32_t)kTriPool) ? askTris : (uint32_t)kTriPool) << 16);
        Closest h; h.t = FAR_T; h.kind = KIND_NONE; h.idx = -1; h.tri = -1; h.u = 0; h.v = 0;
        const bool planesRun = (level == 0) || reflectPlanes;
        if (active) {
            deepest = level;
            cnt.rays++;
            SectionTimer<STATS> tm; traceAnalytic<STATS>(S, ro, rdir, level != 0, planesRun, level != 0, boxesRun, h, cnt.tableFlops); tm.stop(lds, 4);
        }
        // models are traced as primary rays at levels 0 and 2 (kernels.cu:229, 157), as reflection rays at 1 and 3
        if (modelsVisible) {
            const uint32_t nodeTestsBefore = cnt.nodeTests, triTestsBefore = cnt.triTests, nodeVisitsBefore = cnt.nodeVisits;
            tracePool<false, STATS>(S, active, ro, rdir, cull, (level & 1) != 0, P.meshMode, nodeCapacity, (uint32_t)P.traversal, h, lds, cnt);
            if (STATS) {
                if (flags & kCfgCountUseful) {
                    // the same rays once more, bounded from the start by the distance just found: nothing nearer exists, so the
                    // pass finds nothing and only counts what lies in front of the hit
                    Counters again = {};
                    Closest bound; bound.t = h.t; bound.kind = KIND_NONE; bound.idx = -1; bound.tri = -1; bound.u = 0; bound.v = 0;
                    tracePool<false, true, false>(S, active, ro, rdir, cull, (level & 1) != 0, P.meshMode, nodeCapacity, (uint32_t)P.traversal, bound, lds, again);
                    cnt.usefulNodeTests += again.nodeTests; cnt.usefulTriTests += again.triTests; cnt.usefulNodeVisits += again.nodeVisits;
                } else {
                    cnt.usefulNodeTests += cnt.nodeTests - nodeTestsBefore; cnt.usefulTriTests += cnt.triTests - triTestsBefore; cnt.usefulNodeVisits += cnt.nodeVisits - nodeVisitsBefore;
                }
            }
        }
        if (IDS && level == 0 && active) primaryOut = h;
        V3 colour = background;
        bool next = false, lit = false, wantNext = false, add = false, shadowed = false;
        float strength = 0;
        V3 biased = mk(0, 0, 0), toLight = mk(0, 0, 1), point = mk(0, 0, 0), normal = mk(0, 0, 1);
        if (active && h.kind != KIND_NONE) {
            SectionTimer<STATS> tmShade;
            V3 lightNormal = mk(0, 0, 1);
            Surface s = resolveSurface<STATS>(S, ro, rdir, h, renormCount(h.kind, planesRun, boxesRun, modelsRun), cnt, lightNormal);
            const bool isDebug = (s.mat.flags & 1u) != 0;
            if (level == 0 && isDebug) {
                colour = s.mat.ka;                                          // kernels.cu:309-312
            } else {
                V3 nLight;                                                  // kernels.cu:240-245
                if (level == 0 && h.kind == KIND_BOX) nLight = s.normal;
                else if (h.kind == KIND_PLANE) nLight = lightNormal;
                else nLight = normalise(s.normal);
                const V3 lightPos = mk(S.lightPos);
                toLight = normalise(lightPos - s.point);
                colour = lighting(s.mat, lightPos, mk(S.lightCol), s.point, toLight, mk(cam + 13), nLight, P, pixel);
                lit = true;
                point = s.point; normal = s.normal;
                biased = s.point + s.normal * P.shadowBias;
                if (level == 0) wantNext = reflections && s.mat.type == 1u && !(h.kind == KIND_PLANE && !reflectPlanes);
                else wantNext = reflections && s.mat.type == 1u && (1 < P.maxDepth) && level < 3;
                if (wantNext) {
                    if (h.kind == KIND_PLANE) { strength = P.planeStrength; add = true; }
                    else if (h.kind == KIND_SPHERE) { strength = P.sphereStrength; add = true; }
                    else if (h.kind == KIND_MODEL) { strength = P.modelStrength; add = true; }
                    else if (h.kind == KIND_BOX && level == 0) { strength = P.boxStrength; add = true; }   // kernels.cu:259
                }
            }
            tmShade.stop(lds, 5);
        }
        if (level == 0) {
            // shadow rays of the primary hit: the hard shadow ray (kernels.cu:268-277, stops at the first occluder kind), then
            // the soft shadow samples (kernels.cu:279-307, every occluder kind counts separately)
            const int hardRays = hardShadows ? 1 : 0;
            const bool soft = (flags & kCfgSoftShadows) != 0;
            const int softNum = soft ? P.softNum : 0;
            const int shadowRays = hardRays + softNum;
            int hitsCount = 0;
#pragma unroll 1
            for (int q = 0; q < shadowRays; q++) {
                if (__ballot(lit) == 0ull) break;
                const bool hard = q < hardRays;
                V3 sd = toLight;
                bool bySphere = false, byBox = false;
                if (lit) {
                    if (!hard) {
                        V3 ps = lightSampleOffset(sampleSeed(pixelX(pixel), pixelY(pixel), q - hardRays), P.softRadius);
                        V3 lightPoint = mk(S.lightPos) + ps;
                        sd = normalise(lightPoint - point);
                    }
                    cnt.rays++;
                    SectionTimer<STATS> tmSh;
                    bySphere = shadowSpheres<STATS>(S, biased, sd, cnt.tableFlops);
                    byBox = (hard && bySphere) ? false : shadowBoxes<STATS>(S, biased, sd, cnt.tableFlops);
                    tmSh.stop(lds, 6);
                }
                bool byModel = false;
                if (flags & kCfgModelsVisible) {   // the shadow traces ignore renderModels (kernels.cu:304)
                    const bool needModel = lit && !(hard && (bySphere || byBox));
                    Closest dummy; dummy.t = FAR_T; dummy.kind = KIND_NONE; dummy.idx = -1; dummy.tri = -1; dummy.u = 0; dummy.v = 0;
                    const uint32_t nodeTestsBefore = cnt.nodeTests, triTestsBefore = cnt.triTests, nodeVisitsBefore = cnt.nodeVisits;
                    byModel = tracePool<true, STATS>(S, needModel, biased, sd, cull, true, P.meshMode, nodeCapacity, (uint32_t)P.traversal, dummy, lds, cnt);
                    if (STATS) { cnt.usefulNodeTests += cnt.nodeTests - nodeTestsBefore; cnt.usefulTriTests += cnt.triTests - triTestsBefore; cnt.usefulNodeVisits += cnt.nodeVisits - nodeVisitsBefore; }
                }
                if (lit) {
                    if (hard) shadowed = bySphere || byBox || byModel;
                    else hitsCount += (bySphere ? 1 : 0) + (byBox ? 1 : 0) + (byModel ? 1 : 0);
                }
            }
            if (soft && lit) {
                softFactor = 1 - ((float)hitsCount / (float)softNum);
                levelBits |= 256u;
            }
        } else if (hardShadows && lit) {
            cnt.rays++;
            shadowed = shadowSpheres<STATS>(S, biased, toLight, cnt.tableFlops);                                  // kernels.cu:142, 164, 214
        }
        if (lit) {
            if (level == 0) str0 = strength; else if (level == 1) str1 = strength; else if (level == 2) str2 = strength;
            levelBits |= ((shadowed ? 1u : 0u) | (add ? 16u : 0u)) << level;
            if (wantNext) {
                next = true;
                V3 r = normalise(reflect(rdir, normal));
                ro = biased;
                rdir = r;
            }
        }
        if (active) { if (level == 0) lvl0 = colour; else if (level == 1) lvl1 = colour; else if (level == 2) lvl2 = colour; else lvl3 = colour; }
        active = active && next;
    }

    // fold the levels back together in the reference's order of operations
    const float shadowIntensity = P.shadowIntensity;
    V3 c3 = lvl3;
    if (levelBits & 8u) c3 = c3 * shadowIntensity;                               // kernels.cu:140-144
    V3 c2 = lvl2;
    if (levelBits & 4u) c2 = c2 * shadowIntensity;                               // kernels.cu:162-166 (before the add)
    if (deepest >= 3 && (levelBits & 64u)) c2 = c2 + c3 * str2;                      // kernels.cu:168-180
    V3 c1 = lvl1;
    if (deepest >= 2 && (levelBits & 32u)) c1 = c1 + c2 * str1;                      // kernels.cu:198-210
    if (levelBits & 2u) c1 = c1 * shadowIntensity;                               // kernels.cu:212-216 (after the add)
    V3 c0 = lvl0;
    if (deepest >= 1 && (levelBits & 16u)) c0 = c0 + c1 * str0;                      // kernels.cu:248-266
    if (levelBits & 1u) c0 = c0 * shadowIntensity;                               // kernels.cu:268-277
    if (levelBits & 256u) c0 = c0 * softFactor;                                  // kernels.cu:306
    return c0;
}


RT_DEV V3 rasterToCameraSpace(FrameK& P, CamK cam, float x, float y) {
    float ndcx = x / (float)P.width, ndcy = y / (float)P.height;
    float sx = 2 * ndcx - 1, sy = 1 - 2 * ndcy;
    const float fovFactor = cam[12];
    float cx = sx * 1 * fovFactor, cy = sy * fovFactor, cz = 1;
    return (mk(cam + 6) * cx + mk(cam + 3) * cy) + mk(cam + 9) * cz;
}

RT_DEV uint8_t packChannel(float c) {
    float v = selMin(c, 1.0f) * 255;
    if (!(v > 0)) return 0;
    return (uint8_t)f2iRz(v);
}

// One wave per workgroup: a workgroup whose four waves finish at very different times would hold its slots until
// the slowest is done, and wave run times inside one 16x16 block differ by more than 10x at object silhouettes.
constexpr int kWavesPerTileSide = RT_TILE_PX / 8;
constexpr int kWavesPerTile = kWavesPerTileSide * kWavesPerTileSide;

// Bits 0,2,4 of v -> bits 0,1,2 (one coordinate of a Z-order index).
RT_DEV uint32_t evenBits3(uint32_t v) { return (v & 1u) | ((v >> 1) & 2u) | ((v >> 2) & 4u); }

// Regular waves: one wave per 8x8 pixel block, one lane per pixel.  With P.split the grid starts with kMaxExtraWaves split
// waves, the ones listed in fb.extraMap, each rendering one pixel row of one of the frame's heaviest blocks: eight lanes
// bring a pixel, the other 56 only help with the mesh traversal.
// Waves one frame's launch consists of (the regular waves; split waves come on top).
template <typename Frame>
__host__ __device__ inline uint32_t wavesPerFrame(const Frame& P) {
    const int tilesLaunched = P.blockOrder == 2 ? ((P.ownedTiles + 7) / 8) * 8 : P.ownedTiles;
    return (uint32_t)(tilesLaunched * kWavesPerTile);
}

// PATH: one launch renders several frames of one geometry (a camera path), frames x wavesPerFrame workgroups; frame f is
// rendered with the camera pathCams[f] (16 floats: position, up, right, direction, fovFactor, view vector - the layout
// of FrameParams::camPos .. eyeDir) into out + f * pathFrameBytes.  The frames' workgroups are interleaved in dispatch order (see
// the kernel), so the launch has one slow tail, not one per frame.  Nothing is split in this mode.

// The parameter list of renderKernel as it lies in the kernel-argument segment (every parameter at the next offset that
// its alignment allows, which is how a struct with the same members is laid out).
struct RenderArgs {
    DevScene scene; FrameParams frame; uint8_t* out; int32_t* ids; float* tbuf; unsigned long long* counters;
    unsigned long long* waveClocks; const uint32_t* tileOrder; uint32_t* tileCost; BlockFeedback fb;
    const float* pathCams; unsigned long long pathFrameBytes;
};
// Nothing is taken from the parameters themselves: everything is read from the argument segment where it is used (see RT_KARG).
// IDS: the launch also writes the primary hit's ids and / or distance per pixel; without, the six registers of that hit need not
// be kept from the first trace of a pixel to its last.
template <bool STATS, bool PATH = false, bool IDS = true>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_SIMD))) void renderKernel(DevScene, FrameParams, uint8_t*, int32_t*, float*, unsigned long long*,
                                                    unsigned long long*, const uint32_t*, uint32_t*, BlockFeedback, const float*, unsigned long long) {
    // (the address passes through an empty asm statement: a pointer known to be the argument segment is known to be readable
    // everywhere, and every load from it would be moved to the top of the kernel - the very thing this is meant to avoid)
    unsigned long long argSegment = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(argSegment));
    const RT_KARG RenderArgs& A = *(const RT_KARG RenderArgs*)argSegment;
    FrameK& P = A.frame;
    SceneK& S = A.scene;
    uint32_t blockIndex = blockIdx.x, gridBlocks = gridDim.x, frameIndex = 0;
    CamK cam = &P.camPos[0];
    fetchPriority(true);      // until the wave knows its tile (order list, split list: dependent loads) - a new wave gets to its first rays quickly
    if (PATH) {
        // Workgroups are dispatched in index order, and the tile sequence of a frame starts with its most expensive tiles.
        // Frame after frame, the heavy tiles of the last frame would start when the launch is nearly over, and the launch
        // would end with their long waves running alone (eight shares of an eighth of the 1080p frame: 0.32 ms instead of
        // the 0.22 ms the same work takes as one frame).  So the frames are interleaved group by group - a group = the waves
        // of eight tiles, one per XCD, or of one tile without the interleaving over XCDs: group g of every frame comes before
        // group g + 1 of any.
        const uint32_t perFrame = wavesPerFrame(P), frames = gridBlocks / perFrame;
        const uint32_t group = (P.blockOrder == 2 ? 8u : 1u) * (uint32_t)kWavesPerTile;
        const uint32_t slot = blockIndex / group, within = blockIndex - slot * group;
        frameIndex = slot % frames;
        blockIndex = (slot / frames) * group + within; gridBlocks = perFrame;
        cam = (CamK)(unsigned long long)(A.pathCams + 16u * frameIndex);       // wave-uniform: scalar loads
    }
    const unsigned long long clockStart = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz wall clock
    const unsigned long long costStart = __builtin_amdgcn_s_memrealtime();    // costs are wall time in 10 ns units
    if (blockIndex == 0 && threadIdx.x == 0) {
        if (A.fb.hint) *A.fb.hint = *A.fb.extraCount;   // tells the host whether splitting is on
        if (A.fb.frameStart) *A.fb.frameStart = costStart;
    }
    // Grid order with splitting: the regular waves of the heaviest unsplit tiles (the first P.leadWaves waves in tile order,
    // nearly as long as the split blocks), then the split waves, then the other regular waves; workgroups are dispatched in
    // index order.  Share of 1/2, 1/4, 1/8 of the 1080p benchmark frame, one frame in flight: 0.244 / 0.228 / 0.168 ms with
    // the split waves first, 0.187 / 0.162 / 0.142 ms with 2048 regular waves ahead of them, 0.254 / 0.197 / 0.143 ms with
    // all of them ahead; two frames in flight: 0.151 / 0.118 / 0.090 against 0.153 / 0.111 / 0.080 ms.  (Alternating groups
    // of eight split and eight regular waves: 0.273 / 0.211 / 0.160 ms.)
    const uint32_t extraWaves = P.split ? (uint32_t)kMaxExtraWaves : 0u;      // a multiple of 8: the regular waves keep their XCDs
    const uint32_t regularWaves = gridBlocks - extraWaves;
    // ahead of the split waves: the regular waves of the heaviest unsplit tiles (tile order), which are nearly as long
    const uint32_t lead = extraWaves ? ((uint32_t)P.leadWaves < regularWaves ? (uint32_t)P.leadWaves : regularWaves) : 0u;
    const bool splitWave = blockIndex >= lead && blockIndex < lead + extraWaves;
    const uint32_t waveIndex = blockIndex < lead ? blockIndex : blockIndex - (splitWave ? lead : extraWaves);    // among the waves of its kind
    // 10240 bytes per wave: sixteen waves per CU (four per SIMD) fit the 160 KB of LDS exactly.
    __shared__ f4_t poolRays[(RT_RCP_IN_LDS ? 3 : 2) * kBlockThreads];
    __shared__ unsigned long long poolBest[kBlockThreads];
    __shared__ uint32_t poolRec[kBlockThreads];
    __shared__ unsigned long long poolNodes[kNodePool];
    __shared__ unsigned long long poolTris[kTriPool];
    __shared__ unsigned long long profile[16];  // section accumulators of the instrumented kernel: (cycles, executions) x 8
    const uint32_t lane = threadIdx.x;
    WaveLds lds; lds.prof = profile;
    lds.rayO = poolRays; lds.rayD = poolRays + kBlockThreads; lds.rayR = poolRays + 2 * kBlockThreads; lds.bestKey = poolBest; lds.bestRec = poolRec; lds.nodePool = poolNodes; lds.triPool = poolTris;
    if (STATS) { if (lane < 16) profile[lane] = 0; __syncthreads(); }

    // Workgroup B runs on XCD B % 8 (round-robin dispatch).  XCD x takes the tiles at positions x, x + 8, x + 16, ... of the
    // tile sequence: the 16 waves of a tile share one L2, and every XCD gets the same mix of cheap and expensive tiles.
    uint32_t localTile, inTile, sub = 0;
    bool retire = false;
    if (splitWave) {
        const uint32_t count = *inGlobal(A.fb.extraCount);
        if (waveIndex < (count < (uint32_t)kMaxExtraWaves ? count : (uint32_t)kMaxExtraWaves)) {
            const uint32_t e = inGlobal(A.fb.extraMap)[waveIndex];
            const uint32_t blk = e & 0xFFFFFu;
            sub = (e >> 20) & 7u;
            localTile = blk / (uint32_t)kWavesPerTile; inTile = blk % (uint32_t)kWavesPerTile;
            if (localTile >= (uint32_t)P.ownedTiles) retire = true;      // a split list is only ever used for the geometry it was made for; never render outside it
        } else { localTile = 0; inTile = 0; retire = true; }
    } else {
        const uint32_t b = waveIndex;
        if (P.blockOrder == 2) {
            const uint32_t xcd = b % 8u, i = b / 8u;
            localTile = (i / kWavesPerTile) * 8u + xcd;
            inTile = i % kWavesPerTile;
            if (localTile >= (uint32_t)P.ownedTiles) retire = true;
        } else {
            localTile = b / kWavesPerTile; inTile = b % kWavesPerTile;
        }
        // Expensive tiles first: the slowest waves of a frame run several hundred times longer than the median, so they
        // must not be the last to start.  tileOrder lists this rank's tiles by the cost measured on the previous frame.
        if (!retire && A.tileOrder) {
            localTile = inGlobal(A.tileOrder)[localTile];
            // an order list is a permutation of the rank's tiles; should a caller ever hand over one that is not (two launches
            // racing on a shared list), the wave gives up instead of rendering outside the frame
            if (localTile >= (uint32_t)P.ownedTiles) retire = true;
        }
        if (!retire && P.split && *inGlobal(A.fb.extraCount) != 0u) {
            const uint32_t blk = localTile * (uint32_t)kWavesPerTile + inTile;
            if ((inGlobal(A.fb.heavyBits)[blk >> 5] >> (blk & 31u)) & 1u) retire = true;      // split waves render this block
        }
    }
    // diagnostic records: the regular waves first, then the split waves
    const size_t recordIndex = splitWave ? (size_t)regularWaves + waveIndex : (size_t)waveIndex;
    if (retire) {
        if (STATS && A.waveClocks && lane == 0) {
            for (int k = 0; k < kWaveRecordWords; k++) A.waveClocks[kWaveRecordWords * recordIndex + k] = 0;
        }
        return;
    }
    // (the tile came from a vector load of the order list: the same value in every lane, said so here so that everything derived
    // from it is scalar)
    localTile = (uint32_t)__builtin_amdgcn_readfirstlane((int)localTile); inTile = (uint32_t)__builtin_amdgcn_readfirstlane((int)inTile);
    sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
    // (plain interleaving, one tile per period, needs no division)
    const uint32_t run = (uint32_t)P.tileRun, period = run > 1u ? localTile / run : localTile;
    const uint32_t tile = period * (uint32_t)P.world + (uint32_t)P.rank + (localTile - period * run);
    const uint32_t tileX = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile % (uint32_t)P.tilesX)), tileY = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile / (uint32_t)P.tilesX));
    // the waves of a tile walk it in Z order (neighbours in time are neighbours on screen).  A regular wave renders the whole
    // 8x8 block, lane l = pixel (l & 7, l >> 3); a split wave renders row `sub` of the block with its first eight lanes.
    const uint32_t bx = evenBits3(inTile) * 8u, by = evenBits3(inTile >> 1) * 8u + (splitWave ? sub : 0u);
    const bool pixelLane = !splitWave || lane < 8u;
    PixelBase pixel; pixel.x0 = tileX * RT_TILE_PX + bx; pixel.y0 = tileY * RT_TILE_PX + by; pixel.rows = !splitWave;
    const int x = pixelX(pixel), y = pixelY(pixel);

    fetchPriority(false);
    Counters cnt; cnt.tableFlops = 0; cnt.rays = 0; cnt.nodeTests = 0; cnt.triTests = 0; cnt.attrFetches = 0; cnt.gateTests = 0; cnt.nodeVisits = 0;
    cnt.usefulNodeTests = 0; cnt.usefulTriTests = 0; cnt.usefulNodeVisits = 0;
    const bool inFrame = pixelLane && x < P.width && y < P.height;
    {      // all 64 lanes, also those without a pixel: the mesh trace is a wave-wide operation
        const V3 camPos = mk(cam);
        V3 col;
        Closest primary; primary.t = FAR_T; primary.kind = KIND_NONE; primary.idx = -1; primary.tri = -1; primary.u = 0; primary.v = 0;
        // One loop over the camera samples so that the whole ray pipeline is instantiated once: the centre of the pixel
        // (kernels.cu:353-357), or with anti-aliasing the four points (0.25 | 0.75, 0.25 | 0.75), x offset outermost
        // (kernels.cu:331-341).  x + 0.5 is exact in float for any frame width, like the reference's double sum.
        const bool antiAlias = (P.flags & kCfgAntiAlias) != 0;
        const int samples = antiAlias ? 4 : 1;
        V3 total = mk(0, 0, 0);
#pragma unroll 1
        for (int sIdx = 0; sIdx < samples; sIdx++) {
            const float offX = antiAlias ? ((sIdx & 2) ? 0.75f : 0.25f) : 0.5f, offY = antiAlias ? ((sIdx & 1) ? 0.75f : 0.25f) : 0.5f;
            V3 p = rasterToCameraSpace(P, cam, (float)pixelX(pixel) + offX, (float)pixelY(pixel) + offY);
            Closest ph;
            V3 c = castSample<STATS, IDS>(S, P, cam, inFrame, camPos, normalise(p), pixel, lds, cnt, ph);
            if (IDS && sIdx == 0) primary = ph;
            total = antiAlias ? total + c : c;
        }
        col = antiAlias ? total / 4.0f : total;
        if (inFrame) {
            const int xo = pixelX(pixel), yo = pixelY(pixel);      // the lane's pixel, worked out again (see PixelBase)
            size_t pix;
            if (P.compactOutput) pix = (size_t)localTile * (RT_TILE_PX * RT_TILE_PX) + (size_t)((uint32_t)yo % RT_TILE_PX) * RT_TILE_PX + ((uint32_t)xo % RT_TILE_PX);
            else pix = (size_t)yo * (size_t)P.width + (size_t)xo;
            uint8_t* out = inGlobal(A.out);
            if (PATH) out += (size_t)frameIndex * A.pathFrameBytes;
            int32_t* ids = IDS ? inGlobal(A.ids) : nullptr;
            float* tbuf = IDS ? inGlobal(A.tbuf) : nullptr;
            out[pix * 3 + 0] = packChannel(col.x);
            out[pix * 3 + 1] = packChannel(col.y);
            out[pix * 3 + 2] = packChannel(col.z);
            if (ids) {
                size_t fp = (size_t)yo * (size_t)P.width + (size_t)xo;
                bool hit = primary.kind != KIND_NONE;
                ids[fp * 2 + 0] = hit ? ((primary.kind << 24) | (primary.idx & 0xFFFFFF)) : -1;
                // (a model hit holds its position in the scene-wide attribute array; the caller gets the triangle's index in its model)
                int triId = primary.tri;
                if (hit && primary.kind == KIND_MODEL) triId -= (int)inGlobal(S.mdlInfo)[4 * primary.idx + 1];
                ids[fp * 2 + 1] = hit ? triId : -1;
                if (tbuf) tbuf[fp] = primary.t;
            } else if (tbuf) {
                tbuf[(size_t)yo * (size_t)P.width + (size_t)xo] = primary.t;
            }
        }
    }
    if (STATS && A.waveClocks && lane == 0) {      // diagnostic: when this wave started and finished, and its section sums
        unsigned long long* rec = A.waveClocks + kWaveRecordWords * recordIndex;
        rec[0] = clockStart;
        rec[1] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 16; k++) rec[2 + k] = lds.prof[k];
    }
    uint32_t* tileCost = inGlobal(A.tileCost);
    if (tileCost && lane == 0) {
        const unsigned long long spent = __builtin_amdgcn_s_memrealtime() - costStart;
        // a split wave stands for the whole block: count it as what one regular wave would have needed.  The factor is a
        // little below the measured speed-up (1.65) so that a block never looks heavier split than it was unsplit, which
        // would keep a frame that does not profit from splitting in the split state.
        const uint32_t own = (uint32_t)spent + 1u;
        const uint32_t cost = splitWave ? own * 3u / 2u : own;
        atomicMax(&tileCost[localTile], cost);   // a tile is as urgent as its slowest wave
        uint32_t* blockCost = inGlobal(A.fb.blockCost);
        if (blockCost) atomicMax(&blockCost[localTile * (uint32_t)kWavesPerTile + inTile], cost);
    }
    // one atomic per wave and counter
    unsigned long long* counters = inGlobal(A.counters);
    if (counters) {
        uint32_t vals[kCntCount] = {cnt.rays, cnt.nodeTests, cnt.triTests, cnt.attrFetches, cnt.gateTests, cnt.nodeVisits, cnt.tableFlops,
                                    cnt.usefulNodeTests, cnt.usefulTriTests, cnt.usefulNodeVisits};
#pragma unroll
        for (int c = 0; c < kCntCount; c++) {
            if (!STATS && c != kCntRays) continue;
            uint32_t v = vals[c];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && v) atomicAdd(&counters[(blockIdx.x % (uint32_t)kCounterSlots) * (uint32_t)kCounterStride + c], (unsigned long long)v);
        }
    }
}

// Cost bin of a block: four bins per power of two, bin 4e + q holds costs in [(4 + q) << (e - 2), (5 + q) << (e - 2)).
RT_DEV int costBin(uint32_t c) {
    if (c < 4u) return (int)c;
    const int e = 31 - __builtin_clz(c);
    return 4 * e + (int)((c >> (e - 2)) & 3u);
}

// order[rank of tile i by descending cost, ties by index] = i.  One wave per tile, lanes share the n comparisons; used
// for up to kExactRankTiles tiles (2040 at 1080p: 4 us), larger frames are ranked by cost bin (rankTilesBinnedKernel).  Also clears the
// other cost buffer for the frame after next.
__global__ __launch_bounds__(64) void rankTilesKernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order,
                                                      uint32_t* __restrict__ clearNext, int n) {
    const int i = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t mine = cost[i];
    uint32_t rank = 0;
    for (int j = (int)lane; j < n; j += 64) {
        const uint32_t other = cost[j];
        rank += (other > mine || (other == mine && j < i)) ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off, 64);
    if (lane == 0) { order[rank] = (uint32_t)i; clearNext[i] = 0; }
}

// Tile-major buffers of the ranks of a deal laid end to end -> row-major frame.  One thread per 4 output bytes.
// (All three reassembly kernels: blockIdx.y = frame of a batch; frame f reads from gathered + f * frameStrideIn and writes the
// f-th of consecutive frames.)
__global__ __launch_bounds__(256) void untileKernel(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ out, int width, int height,
                                                    TileDeal deal, int tilesX, size_t perRankBytes, size_t frameStrideIn) {
    const size_t total = (size_t)width * (size_t)height * 3;
    gathered += (size_t)blockIdx.y * frameStrideIn; out += (size_t)blockIdx.y * total;
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= total) return;
    uint8_t v[4];
    for (int k = 0; k < 4; k++) {
        size_t i = i4 + k;
        if (i >= total) { v[k] = 0; continue; }
        size_t pix = i / 3; int ch = (int)(i % 3);
        int x = (int)(pix % (size_t)width), y = (int)(pix / (size_t)width);
        int tile = (y / RT_TILE_PX) * tilesX + (x / RT_TILE_PX);
        uint32_t rank, local;
        dealLocate(deal, (uint32_t)tile, rank, local);
        size_t src = (size_t)rank * perRankBytes + ((size_t)local * (RT_TILE_PX * RT_TILE_PX) + (size_t)(y % RT_TILE_PX) * RT_TILE_PX + (x % RT_TILE_PX)) * 3 + ch;
        v[k] = gathered[src];
    }
    if (i4 + 4 <= total) {
        *reinterpret_cast<uint32_t*>(out + i4) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    } else {
        for (int k = 0; k < 4 && i4 + k < total; k++) out[i4 + k] = v[k];
    }
}

// The same for frames whose width is a multiple of 16 pixels: a tile row is 96 contiguous bytes in both layouts and every
// 16-byte piece of the frame lies inside one tile row.  A wave moves kUntilePieces KB of the frame: in every step its 64 lanes
// write 1 KB of the frame in one piece (whole cache lines) and read it as runs of 96 bytes from the tile buffers.  Few, long
// waves: the reassembly runs beside render launches that occupy every wave slot, and what it costs there is the wait for
// slots (1080p on a busy GPU: 26 us per frame with one 16-byte piece per thread, 7 us with six).  Alone on the GPU it moves
// 4.5 TB/s (read + written) at 3840 x 2160; a thread that moved one tile row, six pieces 96 bytes apart from its
// neighbour's, reached 2.2 TB/s.
constexpr uint32_t kUntilePieces = 6;
__global__ __launch_bounds__(256) void untilePiecesKernel(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ out, int width, int height,
                                                          TileDeal deal, int tilesX, size_t perRankBytes, size_t frameStrideIn) {
    const uint32_t piecesPerRow = (uint32_t)width * 3 / 16, pieces = piecesPerRow * (uint32_t)height;
    gathered += (size_t)blockIdx.y * frameStrideIn; out += (size_t)blockIdx.y * (size_t)pieces * 16;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x % 64;
    constexpr uint32_t kTileRowPieces = RT_TILE_PX * 3 / 16;
    u4_t v[kUntilePieces];
#pragma unroll
    for (uint32_t k = 0; k < kUntilePieces; k++) {
        const uint32_t piece = (wave * kUntilePieces + k) * 64 + lane;
        if (piece >= pieces) break;
        const uint32_t y = piece / piecesPerRow, xp = piece - y * piecesPerRow;
        const uint32_t tile = (y / RT_TILE_PX) * (uint32_t)tilesX + xp / kTileRowPieces;
        uint32_t rank, local;
        dealLocate(deal, tile, rank, local);
        const size_t src = (size_t)rank * perRankBytes + ((size_t)local * (RT_TILE_PX * kTileRowPieces) + (size_t)(y % RT_TILE_PX) * kTileRowPieces + xp % kTileRowPieces) * 16;
        v[k] = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(gathered + src));
    }
#pragma unroll
    for (uint32_t k = 0; k < kUntilePieces; k++) {
        const uint32_t piece = (wave * kUntilePieces + k) * 64 + lane;
        if (piece >= pieces) break;
        __builtin_nontemporal_store(v[k], reinterpret_cast<u4_t*>(out) + piece);
    }
}

// ---------------------------------------------------------------- launch wrappers (host)

unsigned launchGridBlocks(const FrameParams& P) { return wavesPerFrame(P); }

hipError_t launchRender(const DevScene& S, const FrameParams& P, uint8_t* out, int32_t* ids, float* tbuf,
                        unsigned long long* counters, bool stats, hipStream_t stream, unsigned long long* waveClocks,
                        const uint32_t* tileOrder, uint32_t* tileCost, BlockFeedback fb) {
    if (P.ownedTiles <= 0) return hipSuccess;
    dim3 grid(launchGridBlocks(P) + (P.split ? (unsigned)kMaxExtraWaves : 0u)), block(kBlockThreads);
    if (stats) hipLaunchKernelGGL((renderKernel<true>), grid, block, 0, stream, S, P, out, ids, tbuf, counters, waveClocks, tileOrder, tileCost, fb, nullptr, 0ull);
    else if (ids || tbuf) hipLaunchKernelGGL((renderKernel<false>), grid, block, 0, stream, S, P, out, ids, tbuf, counters, waveClocks, tileOrder, tileCost, fb, nullptr, 0ull);
    else hipLaunchKernelGGL((renderKernel<false, false, false>), grid, block, 0, stream, S, P, out, ids, tbuf, counters, waveClocks, tileOrder, tileCost, fb, nullptr, 0ull);
    return hipGetLastError();
}

hipError_t launchRenderPath(const DevScene& S, const FrameParams& P, uint8_t* out, size_t frameBytes, int frames, const float* dPathCams,
                            hipStream_t stream, const uint32_t* tileOrder, uint32_t* tileCost) {
    if (P.ownedTiles <= 0 || frames <= 0) return hipSuccess;
    const unsigned long long blocks = (unsigned long long)launchGridBlocks(P) * (unsigned long long)frames;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const BlockFeedback none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((renderKernel<false, true, false>), dim3((unsigned)blocks), dim3(kBlockThreads), 0, stream, S, P, out, (int32_t*)nullptr, (float*)nullptr,
                       (unsigned long long*)nullptr, (unsigned long long*)nullptr, tileOrder, tileCost, none, dPathCams, (unsigned long long)frameBytes);
    return hipGetLastError();
}

// After a frame: sum, maximum and a coarse histogram (cost sums per bin) of the block costs, into slot `slot` of the state.
__global__ __launch_bounds__(256) void blockStatsKernel(const uint32_t* __restrict__ blockCost, uint32_t blocks, SplitState* __restrict__ st, int slot) {
    __shared__ unsigned long long bins[kCostBins];
    if (threadIdx.x < kCostBins) bins[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->used[slot] = 0;      // the selection that follows counts into it
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = i < blocks ? blockCost[i] : 0u;
    if (c) atomicAdd(&bins[costBin(c)], (unsigned long long)c);
    uint32_t m = c;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63u) == 0u && m) atomicMax(&st->maxCost[slot], m);
    __syncthreads();
    if (threadIdx.x < kCostBins && bins[threadIdx.x]) {
        atomicAdd(&st->binSum[slot][threadIdx.x], bins[threadIdx.x]);
        atomicAdd(&st->total[slot], bins[threadIdx.x]);
    }
}

// Chooses the cost bin from which blocks are split in the next frame, and hands those blocks to split waves (eight per
// block, one per pixel row).  Every workgroup derives the same decision from the statistics.
// Model: a split block finishes kSplitGainNum / kSplitGainDen times sooner (measured: 1.65) and its eight waves occupy
// wave slots for 8 x kSplitGainDen / kSplitGainNum times the block's cost; the frame takes max(T, maxCost x Den / Num,
// 1.25 x (sum of costs + extra occupancy) / slots) - the 1.25 is how far a real frame stays behind perfectly even packing.  T runs over the lower
// edges of the top cost bins.  Nothing is split unless the model promises a frame at least 10 % shorter and the frame is
// clearly limited by its heaviest block: that block alone must take more than startTenths / 10 (1.6) times the evenly packed frame to start
// splitting, 0.3 less to carry on (the costs of split blocks are estimates; without the band a frame near the limit
// would flip every other frame).  In practice a GPU that renders a whole 1080p frame is limited by the sum of the work
// and splits nothing, while a rank that renders a fraction of the tiles is limited by its heaviest block and splits a
// few hundred.  ratio256 / 256 scales the weight of the extra occupancy; 0 splits whatever fits (test mode).
// Also clears the block costs and the other slot's statistics.
__global__ __launch_bounds__(256) void selectHeavyKernel(uint32_t* __restrict__ blockCost, uint32_t* __restrict__ extraMap,
                                                        uint32_t* __restrict__ heavyBits, uint32_t blocks, SplitState* __restrict__ st, int slot,
                                                        uint32_t slots, uint32_t ratio256, uint32_t floorCost, uint32_t startTenths) {
    __shared__ int chosenBin;
    if (threadIdx.x == 0) {
        const unsigned long long total = st->total[slot];
        const uint32_t maxCost = st->maxCost[slot];
        const bool wasSplit = st->used[slot ^ 1] != 0u;
        const unsigned long long packed = total * 5u / 4u / (slots ? slots : 1u);
        // the heaviest wave must also span most of the frame as it really ran: with frames of several contexts in flight the
        // waves of this one share the GPU, every cost is stretched, and the frame is not limited by its heaviest block
        const unsigned long long span = __builtin_amdgcn_s_memrealtime() - st->frameStart;
        const bool allow = (unsigned long long)maxCost * 10u > packed * (wasSplit ? startTenths - 3u : startTenths) &&
                           (unsigned long long)maxCost * 10u >= span * 7u;
        const int top = costBin(maxCost);
        int best = top + 1;                                   // nothing split
        if (ratio256 == 0u) best = 0;
        else if (allow && maxCost > 0u) {
            const unsigned long long unsplitTime = maxCost > packed ? maxCost : packed;
            unsigned long long bestTime = unsplitTime, above = 0;
            for (int b = top; b >= 8 && b > top - 14; b--) {
                above += st->binSum[slot][b];
                const int e = b >> 2;
                const unsigned long long edge = (unsigned long long)(4 + (b & 3)) << (e - 2);      // lower edge of bin b
                const unsigned long long extra = above * (8u * kSplitGainDen - kSplitGainNum) / kSplitGainNum;
                const unsigned long long bulk = (total + ((extra * ratio256) >> 8)) * 5u / 4u / (slots ? slots : 1u);
                const unsigned long long splitMax = (unsigned long long)maxCost * kSplitGainDen / kSplitGainNum;
                unsigned long long time = splitMax > edge ? splitMax : edge;
                if (bulk > time) time = bulk;
                if (time < bestTime && time * 10u < unsplitTime * 9u) { bestTime = time; best = b; }
            }
        }
        chosenBin = best;
    }
    if (blockIdx.x == 0) {      // the other slot's statistics are accumulated by the next frame
        if (threadIdx.x < kCostBins) st->binSum[slot ^ 1][threadIdx.x] = 0;
        if (threadIdx.x == 0) { st->total[slot ^ 1] = 0; st->maxCost[slot ^ 1] = 0; }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool heavy = false;
    if (i < blocks) {
        const uint32_t c = blockCost[i];
        blockCost[i] = 0u;
        if (c > 0u && c >= floorCost && costBin(c) >= chosenBin) {
            const uint32_t at = atomicAdd(&st->used[slot], 8u);
            if (at + 8u <= (uint32_t)kMaxExtraWaves) {
                for (uint32_t s = 0; s < 8u; s++) extraMap[at + s] = i | (s << 20);
                heavy = true;
            }
        }
    }
    const unsigned long long mask = __ballot(heavy);
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t word = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
        heavyBits[word] = (uint32_t)mask; heavyBits[word + 1] = (uint32_t)(mask >> 32);
    }
}

hipError_t launchSelectHeavy(uint32_t* blockCost, uint32_t* extraMap, uint32_t* heavyBits, int blocks, SplitState* state, int slot, int slots,
                             int ratio256, uint32_t floorCost, int startTenths, hipStream_t stream) {
    if (blocks <= 0) return hipSuccess;
    const unsigned groups = (unsigned)((blocks + 255) / 256);
    hipLaunchKernelGGL(blockStatsKernel, dim3(groups), dim3(256), 0, stream, blockCost, (uint32_t)blocks, state, slot);
    hipLaunchKernelGGL(selectHeavyKernel, dim3(groups), dim3(256), 0, stream, blockCost, extraMap, heavyBits, (uint32_t)blocks, state, slot,
                       (uint32_t)slots, (uint32_t)ratio256, floorCost, (uint32_t)startTenths);
    return hipGetLastError();
}

int residentRenderWaves() {
    int device = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&device) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, renderKernel<false>, kBlockThreads, 0) != hipSuccess) return 0;
    return cus * perCu;
}

// Ranking of many tiles (the all-pairs kernel needs 0.24 ms for the 32 400 tiles of a 7680 x 4320 frame; one global atomic
// per tile on a few dozen shared counters is no better: same-address atomics from eight XCDs serialise, 0.1 ms for 8160
// tiles).  One workgroup orders the tiles by cost bin in LDS, sixteen bins per power of two, heaviest bin first, in arrival
// order within a bin.
constexpr int kTileBins = 512;
RT_DEV int tileBin(uint32_t c) {
    if (c < 16u) return (int)c;
    const int e = 31 - __builtin_clz(c);
    return 16 * e + (int)((c >> (e - 4)) & 15u);
}

__global__ __launch_bounds__(1024) void rankTilesBinnedKernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order,
                                                             uint32_t* __restrict__ clearNext, int n) {
    __shared__ uint32_t start[kTileBins];
    const int t = (int)threadIdx.x;
    if (t < kTileBins) start[t] = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) atomicAdd(&start[tileBin(cost[i])], 1u);
    __syncthreads();
    if (t < 64) {        // first wave: lane l owns bins [8 (63 - l), 8 (63 - l) + 8), so lane order is descending cost
        const int base = 8 * (63 - t);
        uint32_t cnt[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { cnt[k] = start[base + 7 - k]; sum += cnt[k]; }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (t >= off) incl += v; }
        uint32_t before = incl - sum;
#pragma unroll
        for (int k = 0; k < 8; k++) { start[base + 7 - k] = before; before += cnt[k]; }
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        order[atomicAdd(&start[tileBin(cost[i])], 1u)] = (uint32_t)i;
        clearNext[i] = 0;
    }
}

hipError_t launchRankTiles(const uint32_t* cost, uint32_t* order, uint32_t* clearNext, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (n <= kExactRankTiles) hipLaunchKernelGGL(rankTilesKernel, dim3((unsigned)n), dim3(64), 0, stream, cost, order, clearNext, n);
    else hipLaunchKernelGGL(rankTilesBinnedKernel, dim3(1), dim3(1024), 0, stream, cost, order, clearNext, n);
    return hipGetLastError();
}

// Stream-ordered rewrite of the two table entries the reference's frame loop changes between launches (Window.cpp:357-384):
// sphere 0's geometry and plane 0's ambient colour.  The values arrive by value, so the caller's copies may change at once.
__global__ void patchTablesKernel(F4* sphGeom0, F4 sphere0, F4* plnMat0, float kaX, float kaY, float kaZ) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (sphGeom0) *sphGeom0 = sphere0;
        if (plnMat0) { plnMat0->x = kaX; plnMat0->y = kaY; plnMat0->z = kaZ; }
    }
}

hipError_t launchPatchTables(F4* sphGeom0, F4 sphere0, F4* plnMat0, F4 plane0Mat0, hipStream_t stream) {
    hipLaunchKernelGGL(patchTablesKernel, dim3(1), dim3(64), 0, stream, sphGeom0, sphere0, plnMat0, plane0Mat0.x, plane0Mat0.y, plane0Mat0.z);
    return hipGetLastError();
}

// ---------------------------------------------------------------- scratch that is never allocated
//
// The dispatch feedback of a frame (two tile-cost buffers used alternately, and the tile order derived from them) for callers
// that must not allocate: kStaticTiles entries each, one copy of the array per GPU for the life of the process, shared by
// every context on that GPU.  Host code uses the array only through the per-device address that staticScratch() looks up.
static __device__ __align__(256) __attribute__((used)) uint32_t scratch[3 * kStaticTiles];

namespace {
constexpr int kMaxDevices = 64;
StaticScratch g_staticScratch[kMaxDevices];
std::mutex g_staticScratchLock;
}

hipError_t staticScratch(int device, StaticScratch** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_staticScratchLock);
    StaticScratch& s = g_staticScratch[device];
    if (!s.words) {      // first use on this device: the caller has made `device` current
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        s.words = static_cast<uint32_t*>(p);
        s.costSlot = 0;
        s.orderKey = -1;
    }
    *out = &s;
    return hipSuccess;
}

hipError_t launchUntile(const uint8_t* gathered, uint8_t* out, int width, int height, TileDeal deal, int tilesX, size_t perRankBytes,
                        hipStream_t stream, int frames, size_t frameStrideIn) {
    size_t total = (size_t)width * (size_t)height * 3;
    if (frames <= 0) return hipSuccess;
    static_assert((RT_TILE_PX * 3) % 16 == 0, "a tile row must be a whole number of 16-byte pieces");
    const bool aligned = perRankBytes % 16 == 0 && (reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(out)) % 16 == 0 &&
                         (frames == 1 || (frameStrideIn % 16 == 0 && total % 16 == 0));
    if (aligned && width % 16 == 0 && total / 16 < (1ull << 31)) {
        const size_t waves = (total / 16 + 64 * kUntilePieces - 1) / (64 * kUntilePieces);
        hipLaunchKernelGGL(untilePiecesKernel, dim3((unsigned)((waves + 3) / 4), (unsigned)frames), dim3(256), 0, stream, gathered, out, width, height, deal, tilesX, perRankBytes, frameStrideIn);
        return hipGetLastError();
    }
    size_t threads = (total + 3) / 4;
    dim3 grid((unsigned)((threads + 255) / 256), (unsigned)frames), block(256);
    hipLaunchKernelGGL(untileKernel, grid, block, 0, stream, gathered, out, width, height, deal, tilesX, perRankBytes, frameStrideIn);
    return hipGetLastError();
}

}  // namespace rt
