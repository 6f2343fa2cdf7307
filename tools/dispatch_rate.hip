// Diagnostic: how fast the GPU replaces finished one-wave workgroups that have the render kernel's footprint (64 threads, 128
// vector registers, 10 KB of LDS -> 16 waves per CU).  Each wave spins for a given time on the 100 MHz clock; the launch time
// beyond blocks x duration / 4096 slots is what launching and retiring a wave costs.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void spin(unsigned ticks, unsigned* sink, int loop, unsigned* counter, unsigned total) {
    __shared__ unsigned pad[LDS_BYTES / 4];
    // occupy 128 vector registers
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    pad[threadIdx.x] = threadIdx.x;
    unsigned b = blockIdx.x;
    for (;;) {
        if (loop) {
            unsigned v = 0;
            if (threadIdx.x == 0) v = atomicAdd(counter, 1u);
            b = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            if (b >= total) break;
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
        if (!loop) break;
    }
    if (pad[(threadIdx.x + 1) & 63] == 77777u) sink[0] = b;
}

template <int LDS_BYTES>
static void run(const char* label, unsigned* sink, unsigned* counter) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const unsigned blocks = 32640;
    for (unsigned ticks : {0u, 200u, 500u, 1000u, 2400u}) {     // 0, 2, 5, 10, 24 us
        for (int loop = 0; loop < 2; loop++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipMemsetAsync(counter, 0, 4));
                CHECK(hipEventRecord(a));
                spin<LDS_BYTES><<<loop ? 4096 : blocks, 64>>>(ticks, sink, loop, counter, blocks);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            const double ideal = blocks * (ticks / 100.0) / 4096.0;
            printf("%s  wave %5.1f us  %s: launch %8.1f us, ideal %7.1f us, per wave beyond ideal %6.2f us\n", label, ticks / 100.0,
                   loop ? "4096 looping waves" : "one wave per block", best * 1e3, ideal, (best * 1e3 - ideal) * 4096.0 / blocks);
        }
    }
}

int main() {
    unsigned* sink; unsigned* counter;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&counter, 64));
    run<10240>("10 KB LDS", sink, counter);
    run<256>("256 B LDS", sink, counter);
    return 0;
}
