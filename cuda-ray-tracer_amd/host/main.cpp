// rt_headless: the reference's main() (main.cpp:11-23) without a window.
//   rt_headless [--size WxH] [--preset KEY] [--frames N] [--models DIR] [--out frame.ppm] [--device N]
// Renders the preset's scene N times (default 100, as the reference's render-time average) and prints the timing.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "Headless.h"

int main(int argc, char** argv) {
    int width = 1000, height = 1000, frames = 100, device = 0;   // main.cpp:13 opens a 1000x1000 window
    std::string models = "assets/models", out;
    char preset = 'Z';
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "--size")) { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (!std::strcmp(argv[i], "--preset")) preset = next()[0];
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next());
        else if (!std::strcmp(argv[i], "--models")) models = next();
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next());
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    Headless app(width, height, models, device);
    if (!app.keyPreset(preset)) { std::fprintf(stderr, "unknown preset key '%c' (Z X C V B N M , .)\n", preset); return 2; }
    double avg = app.measure(frames);
    std::printf("Average render time: %f over %d renders (last frame: %.3f ms on the GPU, %llu rays, %.1f Mrays/s)\n", avg, frames,
                app.rayTracer.lastGpuMs, app.rayTracer.lastRayCount,
                app.rayTracer.lastGpuMs > 0 ? app.rayTracer.lastRayCount / (app.rayTracer.lastGpuMs * 1e3) : 0.0);
    if (!out.empty() && !app.writePPM(out)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    return 0;
}
