// rt_view: the reference's main() + Window::run (main.cpp:11-23, Window.cpp:215-326, 338-421) on an X display.
//   rt_view [--size WxH] [--preset KEY] [--models DIR] [--device N] [--frames N]
// Keys as in the reference: Z X C V B N M , . switch scene presets; w a s d, space and left shift move the camera, the
// pointer turns it (left control switches that off and on); i j k l u o move the light; h prints camera and light; r
// restarts the render-time average; Escape quits.  The window title carries what the reference's ImGui panel shows first:
// frames per second, the last render time and the average over 100 renders.  --frames N stops after N frames.
// Exit status 3: no X display.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "Headless.h"
#include "Presenter.h"

int main(int argc, char** argv) {
    int width = 1000, height = 1000, device = 0;      // main.cpp:13 opens a 1000x1000 window
    long frames = 0;
    std::string models = "assets/models";
    char preset = 'Z';
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "--size")) { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (!std::strcmp(argv[i], "--preset")) preset = next()[0];
        else if (!std::strcmp(argv[i], "--models")) models = next();
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atol(next());
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    // the window first: without a display there is nothing to do, and no GPU is touched
    Presenter presenter;
    std::string error;
    if (!presenter.open(width, height, "Ray Tracing", device, error)) {
        std::fprintf(stderr, "rt_view: %s\n", error.c_str());
        return 3;
    }
    Headless app(width, height, models, device);
    app.keyPreset(preset);
    app.rayTracer.renderSharded = [&](rt_ctx* ctx, int w, int h, GLubyte* frame) { return presenter.renderAndPresent(ctx, w, h, frame); };

    const int maxRendersToMeasure = 100;               // Window.cpp:33
    int rendersMeasured = 0, frameCount = 0, fps = 0;
    double totalRenderTimes = 0.0, averageRenderTime = 0.0;
    bool mouseEnabled = true;
    auto savedFrameTime = std::chrono::steady_clock::now();
    Presenter::Input in;
    for (long frame = 0; frames == 0 || frame < frames; frame++) {
        presenter.poll(in);
        if (in.quit) break;
        for (char c : in.pressed) {
            if (c == '^') mouseEnabled = !mouseEnabled;                                   // Window.cpp:64-74
            else if (c == 'r') { rendersMeasured = 0; totalRenderTimes = 0.0; }           // Window.cpp:60-62
            else if (c == 'h') {                                                          // Window.cpp:55-58
                const Scene& s = app.rayTracer.scene;
                const vec3 at = s.cam.getPosition();
                std::printf("%g, %g, %g and %g, %g\n", at.x(), at.y(), at.z(), (double)s.cam.yaw, (double)s.cam.pitch);
                if (!s.lights.empty()) std::printf("Light: %g, %g, %g\n", s.lights[0].position.x(), s.lights[0].position.y(), s.lights[0].position.z());
            } else if (c >= 'a' && c <= 'z') {
                app.keyPreset((char)(c - 'a' + 'A'));
            } else {
                app.keyPreset(c);
            }
        }
        for (char c : in.held) app.keyHeld(c);
        if (mouseEnabled && (in.mouseDx != 0.0 || in.mouseDy != 0.0)) app.rayTracer.scene.cam.mouseMovement(in.mouseDx, in.mouseDy);   // Window.cpp:149-161

        app.display();

        totalRenderTimes += app.lastRenderTime;
        if (++rendersMeasured == maxRendersToMeasure) {                                   // Window.cpp:406-412
            averageRenderTime = totalRenderTimes / rendersMeasured;
            rendersMeasured = 0; totalRenderTimes = 0.0;
        }
        frameCount++;
        const auto now = std::chrono::steady_clock::now();
        if (std::chrono::duration<double>(now - savedFrameTime).count() >= 1.0) {          // Window.cpp:388-394
            fps = frameCount; frameCount = 0; savedFrameTime = now;
            char title[160];
            std::snprintf(title, sizeof title, "Ray Tracing - FPS: %d - render time: %.3f ms - average over %d renders: %.3f ms%s", fps,
                          app.lastRenderTime * 1e3, maxRendersToMeasure, averageRenderTime * 1e3, presenter.interop() ? "" : " (frames through host memory)");
            presenter.setTitle(title);
        }
    }
    app.rayTracer.renderSharded = nullptr;
    presenter.close();
    return 0;
}
