// Checks the render kernel's own math routines (csrc/rt_math.h) on the GPU against the host's libm, the library the CPU
// oracle computes with.  Build and run: see tools/math_check.sh.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../cuda-ray-tracer_amd/csrc/rt_math.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void powKernel(const float* x, const float* y, float* out, float* outLib, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[i] = rt::powSpecular(x[i], y[i]); outLib[i] = powf(x[i], y[i]); }
}

template <int WHICH>
__global__ void powTiming(const float* x, const float* y, float* out, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], acc = 0;
    for (int k = 0; k < reps; k++) { acc += WHICH ? rt::powSpecular(a, b) : powf(a, b); a = a * 0.9999f + 1e-5f; }
    out[i] = acc;
}

static long long ulpDistance(float a, float b) {
    if (std::isnan(a) && std::isnan(b)) return 0;
    if (std::isnan(a) || std::isnan(b)) return 1ll << 40;
    int32_t ia, ib; std::memcpy(&ia, &a, 4); std::memcpy(&ib, &b, 4);
    if (ia < 0) ia = (int32_t)0x80000000 - ia;
    if (ib < 0) ib = (int32_t)0x80000000 - ib;
    return std::llabs((long long)ia - (long long)ib);
}

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    std::vector<float> xs, ys;
    const float shininess[] = {0.0f, 0.25f, 0.5f, 1.0f, 2.0f, 3.0f, 8.0f, 10.0f, 16.0f, 25.0f, 32.0f, 50.0f, 64.0f, 76.8f, 96.0f, 100.0f, 128.0f, 225.0f, 300.0f, 1000.0f, 5000.0f, -1.0f, -7.5f};
    for (float s : shininess) {
        for (int k = 0; k < 400000; k++) {
            double u = uni(rng);
            float x;
            switch (k % 8) {
                case 0: x = (float)u; break;
                case 1: x = (float)(1.0 - u * u * 0.1); break;                 // near 1
                case 2: x = (float)(1.0 - u * 1e-4); break;                    // very near 1
                case 3: x = (float)std::exp2(-u * 40.0); break;                // small
                case 4: x = (float)std::exp2(-126.0 - u * 23.0); break;        // denormal
                case 5: x = (float)(u * 1.5); break;                           // a little above 1 as well
                case 6: x = (float)std::sqrt(u); break;
                default: x = (float)(0.5 + 0.5 * u); break;
            }
            xs.push_back(x); ys.push_back(s);
        }
    }
    for (int k = 0; k < 2000000; k++) { xs.push_back((float)uni(rng)); ys.push_back((float)(uni(rng) * 200.0)); }
    const float special[] = {0.0f, 1.0f, -0.0f, -1.0f, 2.0f, INFINITY, -INFINITY, NAN, 1e-45f, 3.4e38f, 0.5f, -0.5f, 1e30f, -3.0f, 4.0f};
    for (float a : special) for (float b : special) { xs.push_back(a); ys.push_back(b); }
    const int n = (int)xs.size();
    float *dx, *dy, *dout, *dlib;
    CHECK(hipMalloc(&dx, n * 4)); CHECK(hipMalloc(&dy, n * 4)); CHECK(hipMalloc(&dout, n * 4)); CHECK(hipMalloc(&dlib, n * 4));
    CHECK(hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dy, ys.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(powKernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, dout, dlib, n);
    CHECK(hipDeviceSynchronize());
    std::vector<float> out(n), lib(n);
    CHECK(hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(lib.data(), dlib, n * 4, hipMemcpyDeviceToHost));
    long long worstOwn = 0, worstLib = 0, offOwn = 0, offLib = 0, offOwnRounded = 0, offLibRounded = 0, specialBad = 0;
    int worstAt = -1;
    for (int i = 0; i < n; i++) {
        const float host = powf(xs[i], ys[i]);                       // what the oracle computes
        const float rounded = (float)std::pow((double)xs[i], (double)ys[i]);    // the double result rounded once
        const long long a = ulpDistance(out[i], host), b = ulpDistance(lib[i], host);
        if (a > worstOwn) { worstOwn = a; worstAt = i; }
        if (b > worstLib) worstLib = b;
        offOwn += a != 0; offLib += b != 0;
        offOwnRounded += ulpDistance(out[i], rounded) != 0; offLibRounded += ulpDistance(lib[i], rounded) != 0;
        if (i >= n - (int)(sizeof(special) / sizeof(float)) * (int)(sizeof(special) / sizeof(float)) && a != 0) {
            specialBad++;
            std::printf("special case pow(%g, %g): own %g host %g\n", xs[i], ys[i], out[i], host);
        }
    }
    std::printf("pow: %d cases; against the host's powf: own routine differs in %lld (worst %lld ulp, at pow(%.9g, %.9g) = %.9g vs %.9g), device library in %lld (worst %lld ulp)\n",
                n, offOwn, worstOwn, xs[worstAt], ys[worstAt], out[worstAt], powf(xs[worstAt], ys[worstAt]), offLib, worstLib);
    std::printf("pow: against the double result rounded once: own routine differs in %lld, device library in %lld\n", offOwnRounded, offLibRounded);
    // timing
    const int tn = 1 << 22, reps = 64;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipEventRecord(e0));
            if (which) hipLaunchKernelGGL(powTiming<1>, dim3(tn / 256), dim3(256), 0, 0, dx, dy, dout, tn, reps);
            else hipLaunchKernelGGL(powTiming<0>, dim3(tn / 256), dim3(256), 0, 0, dx, dy, dout, tn, reps);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::printf("pow timing: %s %.3f ms for %d x %d evaluations\n", which ? "own routine" : "device library", ms, tn, reps);
    }
    return (worstOwn <= 1 && specialBad == 0) ? 0 : 1;
}
