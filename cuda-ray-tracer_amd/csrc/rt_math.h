// Device math routines of the render kernel that replace library calls (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// x^y for the specular term (kernels.cu:48-57: pow(max(r.v, 0), shininess)), evaluated through double precision:
// log2 x = e + (2 / ln 2) atanh((m - 1) / (m + 1)) with x = m 2^e, m in [sqrt(1/2), sqrt(2)), then 2^(y log2 x) by a
// polynomial on the fractional part.  Both polynomials are good to 2^-34 or better, so the double result is rounded to the
// nearest float except when it lies within about 2^-10 of a float's half-way point, where it may be the neighbour: closer to
// the correctly rounded value than the device library's powf (under one unit in the last place), at a third of its
// instructions - double-precision fma, multiply and add run at the full vector rate on this GPU.
// The fast path covers +0 <= x < inf with |y| < 2^40; everything else (negative, -0, infinite or NaN arguments) takes powf.
// Wave-uniform control flow: the library call is skipped only if no active lane needs it.
// Not inlined: inside the shading code its double-precision temporaries cost the render kernel 38 more spilled registers and
// 7 % of the frame; as a function it has registers of its own.
__device__ __noinline__ float powSpecular(float x, float y) {
    const bool fast = __float_as_uint(x) < 0x7f800000u && __builtin_fabsf(y) < 1099511627776.0f;
    float result = 0.0f;
    if (__ballot(!fast) != 0ull) {
        if (!fast) result = powf(x, y);
    }
    if (fast) {
        const double xd = (double)x;
        double m = __builtin_amdgcn_frexp_mant(xd);              // [0.5, 1)
        int e = __builtin_amdgcn_frexp_exp(xd);
        const bool low = m < 0.70710678118654752440;
        m = low ? m * 2.0 : m;
        e = low ? e - 1 : e;
        const double f = m - 1.0, d = m + 1.0;                   // d in [1.70, 2.42)
        const double r0 = (double)__builtin_amdgcn_rcpf((float)d);
        const double r = __builtin_fma(__builtin_fma(-d, r0, 1.0), r0, r0);
        const double s = f * r, s2 = s * s;
        // atanh(s) / s in s^2 on [0, 0.0295], times 2 / ln 2
        double p = 0.0980530622568066;
        p = __builtin_fma(p, s2, 0.1108703396891863);
        p = __builtin_fma(p, s2, 0.14286085439308532);
        p = __builtin_fma(p, s2, 0.19999997419574225);
        p = __builtin_fma(p, s2, 0.3333333333990304);
        p = __builtin_fma(p, s2, 0.9999999999999726);
        const double log2x = __builtin_fma(s * 2.8853900817779268147, p, (double)e);
        double z = (double)y * log2x;
        z = __builtin_fmin(__builtin_fmax(z, -200.0), 200.0);    // beyond: 0 or inf after the conversion to float
        const double n = __builtin_rint(z), t = z - n;           // t in [-0.5, 0.5]
        double q = 1.5303699001378845e-05;
        q = __builtin_fma(q, t, 0.00015469729140704658);
        q = __builtin_fma(q, t, 0.0013333478480888061);
        q = __builtin_fma(q, t, 0.009618025613598036);
        q = __builtin_fma(q, t, 0.055504109063177154);
        q = __builtin_fma(q, t, 0.24022651213495422);
        q = __builtin_fma(q, t, 0.6931471805568341);
        q = __builtin_fma(q, t, 0.9999999999595622);
        float v = (float)__builtin_amdgcn_ldexp(q, (int)n);
        // pow(+0, y): 0 for y > 0, 1 for y = 0, inf for y < 0; pow(x, 0) = 1 falls out of z = 0
        if (x == 0.0f) v = y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());
        result = v;
    }
    return result;
}

}  // namespace rt
