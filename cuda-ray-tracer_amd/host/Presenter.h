// On-screen presentation of the rendered frame: what the reference's Window + Square do with GLFW, GLEW and a texture
// upload from managed memory (Window.cpp:164-213, 338-421; Square.cpp:16-19, 82-94), done with Xlib + GLX and HIP's OpenGL
// interop.  The frame is rendered by rt_render_device straight into a pixel buffer object that OpenGL owns; the texture is
// then filled from that buffer on the GPU, so a frame never crosses PCIe.  If the OpenGL context does not live on the GPU
// that renders (remote display, software rasteriser), the presenter falls back to the reference's route: render to the
// host framebuffer and upload it with glTexSubImage2D.
//
// Part of rt_view only; nothing else links X11 or OpenGL.
#pragma once
#include <string>
#include <vector>

#include "rt_api.h"

class Presenter {
public:
    Presenter();
    ~Presenter();
    Presenter(const Presenter&) = delete;
    Presenter& operator=(const Presenter&) = delete;

    // Opens a width x height window on $DISPLAY with an OpenGL context, creates the texture and the pixel buffer and
    // registers the buffer with HIP on `device`.  Returns false with a message in `error` if there is no display or no
    // usable visual; a context on another GPU is not an error (see interop()).
    bool open(int width, int height, const char* title, int device, std::string& error);
    void close();
    bool isOpen() const { return impl_ != nullptr; }
    // true: frames are rendered into the OpenGL buffer on the GPU; false: through `hostFrame` and a texture upload
    bool interop() const;

    // Renders the prepared frame of `ctx` (scene and frame state already set) and shows it.  `hostFrame` (width * height * 3
    // bytes) is only written on the fallback route.  Returns an rt_status.
    int renderAndPresent(rt_ctx* ctx, int width, int height, unsigned char* hostFrame);

    struct Input {
        std::vector<char> pressed;      // keys that went down since the last poll, as the characters Window::keyInput uses
        std::vector<char> held;         // movement keys that are down now: w a s d _ (space) | (left shift) i j k l u o
        double mouseDx = 0.0, mouseDy = 0.0;   // pointer movement since the last poll (Window::mouseInput)
        bool quit = false;              // Escape or the window was closed
    };
    // Drains the X event queue.
    void poll(Input& in);
    void setTitle(const std::string& title);

private:
    struct Impl;
    Impl* impl_ = nullptr;
};
