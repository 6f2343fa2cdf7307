// Headless stand-in for the reference's Window (Window.h, Window.cpp): no GLFW, no OpenGL, no ImGui.  It keeps the
// parts of Window that define what is rendered and how it is timed: the key presets (Window.cpp:76-145), the
// per-frame fix-ups (Window.cpp:382-384), the light/camera key moves (Window.cpp:339-380) and the render-time
// average over N launches (Window.cpp:400-412).  Frames are written as binary PPM instead of being drawn.
#pragma once
#include <string>

#include "RayTracer.h"

class Headless {
public:
    RayTracer rayTracer;
    double lastRenderTime = 0.0;        // seconds, host wall clock around launchKernel (Window.cpp:400-403)
    double averageRenderTime = 0.0;

    Headless(int width, int height, const std::string& modelDirectory, int device = 0);

    // A scene/preset key of the reference: Z X C V B N M , .   Returns false for any other key.
    bool keyPreset(char key);
    // A held movement key for one frame: w a s d _ | (camera), i j k l u o (light and its marker sphere).
    void keyHeld(char key);
    // One iteration of Window::display: fix-ups, render, timing.
    void display();
    // Renders `frames` frames and returns the average render time in seconds (Window.cpp:406-412).
    double measure(int frames);
    // The frame loop with input (Window::run + Window::display, Window.cpp:215-326, 338-421): before frame i the key keys[i % n]
    // is held for one frame (see keyHeld; '<' and '>' turn the camera by 20 mouse units, ' ' holds nothing), then the frame is
    // rendered; with a directory, every frame is written there as frame_0000.ppm, frame_0001.ppm, ...  Returns the average
    // render time in seconds, or a negative value if a frame could not be written.
    double runSequence(int frames, const std::string& keys, const std::string& directory);

    bool writePPM(const std::string& path) const;
};
