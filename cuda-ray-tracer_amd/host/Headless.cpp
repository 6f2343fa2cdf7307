#include "Headless.h"

#include <chrono>
#include <cstdio>

Headless::Headless(int width, int height, const std::string& modelDirectory, int device) {
    rayTracer.modelDirectory = modelDirectory;
    rayTracer.device = device;
    rayTracer.init(width, height);
    rayTracer.scene.cam.updateDirection();      // Window.cpp:193
}

bool Headless::keyPreset(char key) {
    SceneConfig& c = rayTracer.config;
    switch (key) {
        case 'Z': rayTracer.switchScene(0); break;
        case 'X':
            rayTracer.switchScene(1);
            c.maxDepth = 1; c.renderHardShadows = false; c.renderSoftShadows = false; c.reflections = true; c.reflectPlanes = true;
            c.planeReflectionStrength = 0.5f; c.sphereReflectionStrength = 0.7f;
            break;
        case 'C': rayTracer.switchScene(5); c.reflections = false; c.renderHardShadows = true; break;
        case 'V': rayTracer.switchScene(2); c.renderHardShadows = false; c.renderSoftShadows = false; c.reflections = false; break;
        case 'B':
            rayTracer.switchScene(3);
            c.reflections = true; c.planeReflectionStrength = 0.22f; c.renderHardShadows = false; c.renderSoftShadows = false;
            break;
        case 'N':
            rayTracer.switchScene(7);
            c.renderHardShadows = false; c.renderSoftShadows = false; c.reflections = true; c.reflectPlanes = true; c.maxDepth = 1;
            c.planeReflectionStrength = 0.5f;
            break;
        case 'M': c.renderHardShadows = false; c.renderSoftShadows = false; c.reflections = false; rayTracer.switchScene(6); break;
        case ',': rayTracer.switchScene(8); c.reflections = true; c.reflectPlanes = true; c.planeReflectionStrength = 0.5f; break;
        case '.':
            rayTracer.switchScene(4);
            c.renderHardShadows = true; c.renderSoftShadows = false; c.reflections = true; c.modelReflectionStrength = 0.8f;
            break;
        default: return false;
    }
    rayTracer.scene.cam.mouseMovement(0, 0);    // every preset ends with this (e.g. Window.cpp:78)
    return true;
}

void Headless::keyHeld(char key) {
    Scene& s = rayTracer.scene;
    auto moveLight = [&](int axis, float by) {
        if (s.lights.empty() || s.spheres.empty()) return;
        s.lights[0].position[axis] += by;
        s.spheres[0].position[axis] += by;
    };
    switch (key) {
        case 'w': s.cam.move(Camera::FORWARD); break;
        case 's': s.cam.move(Camera::BACKWARD); break;
        case 'a': s.cam.move(Camera::LEFT); break;
        case 'd': s.cam.move(Camera::RIGHT); break;
        case '_': s.cam.move(Camera::UP); break;
        case '|': s.cam.move(Camera::DOWN); break;
        case 'i': moveLight(2, -1); break;
        case 'k': moveLight(2, 1); break;
        case 'j': moveLight(0, -1); break;
        case 'l': moveLight(0, 1); break;
        case 'u': moveLight(1, 1); break;
        case 'o': moveLight(1, -1); break;
        default: break;
    }
}

void Headless::display() {
    Scene& s = rayTracer.scene;
    SceneConfig& c = rayTracer.config;
    if (!s.spheres.empty()) s.spheres[0].radius = (float)c.softShadowRadius;
    if (!s.planes.empty()) s.planes[0].mat.ambientColour = vec3(0.1f, 0.1f, 0.1f) * (float)c.floorBrightness;
    c.backgroundCol = vec3(0.1f, 0.1f, 0.1f) * (float)c.backgroundBrightness;

    auto before = std::chrono::steady_clock::now();
    rayTracer.launchKernel();
    auto after = std::chrono::steady_clock::now();
    lastRenderTime = std::chrono::duration<double>(after - before).count();
}

double Headless::measure(int frames) {
    double total = 0.0;
    for (int i = 0; i < frames; i++) { display(); total += lastRenderTime; }
    averageRenderTime = frames > 0 ? total / frames : 0.0;
    return averageRenderTime;
}

double Headless::runSequence(int frames, const std::string& keys, const std::string& directory) {
    double total = 0.0;
    for (int i = 0; i < frames; i++) {
        if (!keys.empty()) {
            const char k = keys[(size_t)i % keys.size()];
            if (k == '<') rayTracer.scene.cam.mouseMovement(-20.0, 0.0);
            else if (k == '>') rayTracer.scene.cam.mouseMovement(20.0, 0.0);
            else if (k != ' ') keyHeld(k);
        }
        display();
        total += lastRenderTime;
        if (!directory.empty()) {
            char name[32];
            std::snprintf(name, sizeof name, "/frame_%04d.ppm", i);
            if (!writePPM(directory + name)) return -1.0;
        }
    }
    averageRenderTime = frames > 0 ? total / frames : 0.0;
    return averageRenderTime;
}

bool Headless::writePPM(const std::string& path) const {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", rayTracer.width, rayTracer.height);
    size_t n = (size_t)rayTracer.width * (size_t)rayTracer.height * 3;
    bool ok = std::fwrite(rayTracer.framebuffer, 1, n, f) == n;
    std::fclose(f);
    return ok;
}
