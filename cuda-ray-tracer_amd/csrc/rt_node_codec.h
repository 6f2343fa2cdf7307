// Encoder of the 128-byte octree node record (layout in rt_device_types.h), shared by the host flattener and the
// device octree builder so that both produce bit-identical, outward-rounded child boxes.
#pragma once
#include <math.h>
#include <stdint.h>

#include "rt_device_types.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt {

RT_HD inline float codecBitsToFloat(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

RT_HD inline uint32_t codecFloatToBits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

// Encodes one node record into 32 words: the eight child boxes lo/hi[axis][slot] (already padded; ignored for slots in
// neither mask) are quantised outwards against the node's own origin and power-of-two steps.  Integer words on
// purpose: references and masks are small integers, which as floats would be denormals.
RT_HD inline void encodeNodeWords(uint32_t words[32], const float lo[3][8], const float hi[3][8], const uint32_t leafRef[8],
                                  uint32_t childBase, uint32_t imask, uint32_t lmask, uint32_t model) {
    const uint32_t used = imask | lmask;
    float origin[3] = {0.0f, 0.0f, 0.0f};
    uint32_t expo[3] = {127u, 127u, 127u};
    uint32_t planes[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int a = 0; a < 3; a++) {
        double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
        for (int c = 0; c < 8; c++) {
            if (!((used >> c) & 1u)) continue;
            mn = fmin(mn, (double)lo[a][c]); mx = fmax(mx, (double)hi[a][c]);
        }
        double step = 1.0;
        if (used) {
            origin[a] = (float)mn;
            int ex = -126;
            if (mx > mn) { (void)frexp((mx - mn) / 255.0, &ex); }     // 2^ex >= extent / 255
            ex = ex > 127 ? 127 : ex < -126 ? -126 : ex;
            expo[a] = (uint32_t)(ex + 127);
            step = ldexp(1.0, ex);
        }
        for (int c = 0; c < 8; c++) {
            uint32_t ql = 255u, qh = 0u;
            if ((used >> c) & 1u) {
                ql = (uint32_t)fmin(255.0, fmax(0.0, floor(((double)lo[a][c] - (double)origin[a]) / step)));
                qh = (uint32_t)fmin(255.0, fmax(0.0, ceil(((double)hi[a][c] - (double)origin[a]) / step)));
            }
            planes[a][c / 4] |= ql << (8 * (c % 4));
            planes[a][2 + c / 4] |= qh << (8 * (c % 4));
        }
    }
    words[0] = codecFloatToBits(origin[0]); words[1] = codecFloatToBits(origin[1]); words[2] = codecFloatToBits(origin[2]);
    words[3] = expo[0] | (expo[1] << 8) | (expo[2] << 16);
    words[4] = childBase; words[5] = imask | (lmask << 8); words[6] = model; words[7] = 0u;
    for (int a = 0; a < 3; a++) for (int k = 0; k < 4; k++) words[8 + 4 * a + k] = planes[a][k];
    for (int c = 0; c < 8; c++) words[20 + c] = leafRef[c];
    for (int k = 28; k < 32; k++) words[k] = 0u;
}

RT_HD inline void encodeNode(F4* rec, const float lo[3][8], const float hi[3][8], const uint32_t leafRef[8], uint32_t childBase,
                             uint32_t imask, uint32_t lmask, uint32_t model) {
    uint32_t words[32];
    encodeNodeWords(words, lo, hi, leafRef, childBase, imask, lmask, model);
    __builtin_memcpy(rec, words, sizeof words);
}

// Triangle id word of the leaf-ordered triangle array: index inside the model, chunk of the eight-box mode, whether the
// triangle sits in the chunk's never-tested last slot (Model.cpp:229: j < indicesCount - 3), last-in-leaf flag.
// perBoxTris = ceil((indicesCount / 3) / 8) (Model.cpp:62); chunkIndices[b] = boundingBoxes[b].indicesCount.
RT_HD inline uint32_t chunkTagOf(int64_t t, int64_t perBoxTris, const int32_t chunkIndices[8]) {
    const int64_t b = perBoxTris > 0 ? t / perBoxTris : 0, slot = perBoxTris > 0 ? t % perBoxTris : 0;
    if (b >= 8) return 0u;
    const bool lastSlot = 3 * (slot + 1) >= (int64_t)chunkIndices[b];
    return ((uint32_t)b << kTriChunkShift) | (lastSlot ? kTriChunkDropped : 0u);
}

}  // namespace rt
