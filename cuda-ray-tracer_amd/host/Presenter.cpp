#include "Presenter.h"

// HIP first: Xlib defines None, Bool, Status and friends as macros.
#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/glx.h>
#include <X11/Xlib.h>
#include <X11/Xutil.h>
#include <X11/keysym.h>

#include <cstdio>
#include <cstring>

struct Presenter::Impl {
    Display* display = nullptr;
    ::Window window = 0;
    Colormap colormap = 0;
    GLXContext context = nullptr;
    Atom deleteMessage = 0;
    int width = 0, height = 0, device = 0;
    GLuint texture = 0, pixelBuffer = 0;
    hipGraphicsResource* resource = nullptr;
    hipStream_t stream = nullptr;
    bool interop = false;
    bool down[256] = {};
    bool havePointer = false;
    int pointerX = 0, pointerY = 0;
    PFNGLGENBUFFERSPROC genBuffers = nullptr;
    PFNGLBINDBUFFERPROC bindBuffer = nullptr;
    PFNGLBUFFERDATAPROC bufferData = nullptr;
    PFNGLDELETEBUFFERSPROC deleteBuffers = nullptr;
};

namespace {

template <typename F>
F glFunction(const char* name) { return reinterpret_cast<F>(glXGetProcAddressARB(reinterpret_cast<const GLubyte*>(name))); }

// The character Window::keyInput files a key under (Window.cpp:42-53, 76-145), 0 for keys it ignores.
char keyCharacter(KeySym sym) {
    if (sym >= XK_a && sym <= XK_z) return (char)('a' + (sym - XK_a));
    if (sym >= XK_A && sym <= XK_Z) return (char)('a' + (sym - XK_A));
    switch (sym) {
        case XK_space: return '_';
        case XK_Shift_L: return '|';
        case XK_comma: return ',';
        case XK_period: return '.';
        case XK_Control_L: return '^';
        default: return 0;
    }
}

}  // namespace

Presenter::Presenter() {}
Presenter::~Presenter() { close(); }

bool Presenter::interop() const { return impl_ && impl_->interop; }

bool Presenter::open(int width, int height, const char* title, int device, std::string& error) {
    close();
    if (width <= 0 || height <= 0) { error = "window size must be positive"; return false; }
    Impl* p = new Impl;
    p->width = width; p->height = height; p->device = device;
    p->display = XOpenDisplay(nullptr);
    if (!p->display) { delete p; error = "cannot open the X display (is DISPLAY set?)"; return false; }
    impl_ = p;
    const int screen = DefaultScreen(p->display);
    int attributes[] = {GLX_RGBA, GLX_DOUBLEBUFFER, GLX_RED_SIZE, 8, GLX_GREEN_SIZE, 8, GLX_BLUE_SIZE, 8, 0};
    XVisualInfo* visual = glXChooseVisual(p->display, screen, attributes);
    if (!visual) { close(); error = "the X server offers no double-buffered RGB visual with OpenGL"; return false; }
    const ::Window root = RootWindow(p->display, visual->screen);
    p->colormap = XCreateColormap(p->display, root, visual->visual, AllocNone);
    XSetWindowAttributes wa;
    std::memset(&wa, 0, sizeof wa);
    wa.colormap = p->colormap;
    wa.event_mask = KeyPressMask | KeyReleaseMask | PointerMotionMask | StructureNotifyMask;
    p->window = XCreateWindow(p->display, root, 0, 0, (unsigned)width, (unsigned)height, 0, visual->depth, InputOutput, visual->visual,
                              CWColormap | CWEventMask, &wa);
    XStoreName(p->display, p->window, title ? title : "");
    p->deleteMessage = XInternAtom(p->display, "WM_DELETE_WINDOW", False);
    XSetWMProtocols(p->display, p->window, &p->deleteMessage, 1);
    XMapWindow(p->display, p->window);
    p->context = glXCreateContext(p->display, visual, nullptr, True);
    XFree(visual);
    if (!p->context || !glXMakeCurrent(p->display, p->window, p->context)) { close(); error = "cannot create an OpenGL context"; return false; }

    // the texture the frame is shown through (Square.cpp:28-34)
    glClearColor(0.0f, 0.0f, 0.0f, 0.0f);
    glGenTextures(1, &p->texture);
    glBindTexture(GL_TEXTURE_2D, p->texture);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);      // rows of width * 3 bytes, any width
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB, width, height, 0, GL_RGB, GL_UNSIGNED_BYTE, nullptr);

    // The buffer the kernel renders into.  Everything from here on may fail without failing open(): the presenter then
    // shows frames that came through host memory.
    p->genBuffers = glFunction<PFNGLGENBUFFERSPROC>("glGenBuffers");
    p->bindBuffer = glFunction<PFNGLBINDBUFFERPROC>("glBindBuffer");
    p->bufferData = glFunction<PFNGLBUFFERDATAPROC>("glBufferData");
    p->deleteBuffers = glFunction<PFNGLDELETEBUFFERSPROC>("glDeleteBuffers");
    if (p->genBuffers && p->bindBuffer && p->bufferData && p->deleteBuffers && hipSetDevice(device) == hipSuccess) {
        unsigned int count = 0;
        int devices[16];
        bool sameGpu = false;
        if (hipGLGetDevices(&count, devices, 16, hipGLDeviceListAll) == hipSuccess)
            for (unsigned int i = 0; i < count && i < 16; i++) sameGpu = sameGpu || devices[i] == device;
        if (sameGpu) {
            p->genBuffers(1, &p->pixelBuffer);
            p->bindBuffer(GL_PIXEL_UNPACK_BUFFER, p->pixelBuffer);
            p->bufferData(GL_PIXEL_UNPACK_BUFFER, (GLsizeiptr)width * height * 3, nullptr, GL_STREAM_DRAW);
            p->bindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
            if (glGetError() == GL_NO_ERROR &&
                hipGraphicsGLRegisterBuffer(&p->resource, p->pixelBuffer, hipGraphicsRegisterFlagsWriteDiscard) == hipSuccess &&
                hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess) {
                p->interop = true;
            } else {
                (void)hipGetLastError();
                if (p->resource) { (void)hipGraphicsUnregisterResource(p->resource); p->resource = nullptr; }
            }
        }
    }
    if (!p->interop) std::fprintf(stderr, "Presenter: the OpenGL context is not on GPU %d; frames go through host memory\n", device);
    return true;
}

void Presenter::close() {
    Impl* p = impl_;
    if (!p) return;
    if (p->resource) (void)hipGraphicsUnregisterResource(p->resource);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->display) {
        if (p->context) {
            if (p->pixelBuffer && p->deleteBuffers) p->deleteBuffers(1, &p->pixelBuffer);
            if (p->texture) glDeleteTextures(1, &p->texture);
            glXMakeCurrent(p->display, 0, nullptr);
            glXDestroyContext(p->display, p->context);
        }
        if (p->window) XDestroyWindow(p->display, p->window);
        if (p->colormap) XFreeColormap(p->display, p->colormap);
        XCloseDisplay(p->display);
    }
    delete p;
    impl_ = nullptr;
}

int Presenter::renderAndPresent(rt_ctx* ctx, int width, int height, unsigned char* hostFrame) {
    Impl* p = impl_;
    if (!p || !ctx) return RT_ERR_INVALID;
    if (width != p->width || height != p->height) return RT_ERR_INVALID;      // the window's texture and buffer have one size
    glBindTexture(GL_TEXTURE_2D, p->texture);
    if (p->interop) {
        if (hipSetDevice(p->device) != hipSuccess || hipGraphicsMapResources(1, &p->resource, p->stream) != hipSuccess) return RT_ERR_DEVICE;
        void* mapped = nullptr;
        size_t bytes = 0;
        int rc = RT_ERR_DEVICE;
        if (hipGraphicsResourceGetMappedPointer(&mapped, &bytes, p->resource) == hipSuccess && bytes >= (size_t)width * (size_t)height * 3)
            rc = rt_render_device(ctx, width, height, mapped, p->stream);
        // OpenGL may read the buffer as soon as it is unmapped
        if (hipStreamSynchronize(p->stream) != hipSuccess && rc == RT_OK) rc = RT_ERR_DEVICE;
        if (hipGraphicsUnmapResources(1, &p->resource, p->stream) != hipSuccess && rc == RT_OK) rc = RT_ERR_DEVICE;
        if (rc != RT_OK) return rc;
        p->bindBuffer(GL_PIXEL_UNPACK_BUFFER, p->pixelBuffer);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, width, height, GL_RGB, GL_UNSIGNED_BYTE, nullptr);    // offset 0 of the bound buffer
        p->bindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
    } else {
        if (!hostFrame) return RT_ERR_INVALID;
        const int rc = rt_render_frame(ctx, width, height, hostFrame, nullptr);
        if (rc != RT_OK) return rc;
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, width, height, GL_RGB, GL_UNSIGNED_BYTE, hostFrame);    // Square.cpp:16-19
    }
    // two triangles over the whole window, texture row 0 at the bottom (Square.cpp:38-54)
    glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    glEnable(GL_TEXTURE_2D);
    glTexEnvi(GL_TEXTURE_ENV, GL_TEXTURE_ENV_MODE, GL_REPLACE);
    glBegin(GL_TRIANGLES);
    glTexCoord2f(0.0f, 1.0f); glVertex3f(-1.0f, 1.0f, 0.0f);
    glTexCoord2f(0.0f, 0.0f); glVertex3f(-1.0f, -1.0f, 0.0f);
    glTexCoord2f(1.0f, 0.0f); glVertex3f(1.0f, -1.0f, 0.0f);
    glTexCoord2f(0.0f, 1.0f); glVertex3f(-1.0f, 1.0f, 0.0f);
    glTexCoord2f(1.0f, 1.0f); glVertex3f(1.0f, 1.0f, 0.0f);
    glTexCoord2f(1.0f, 0.0f); glVertex3f(1.0f, -1.0f, 0.0f);
    glEnd();
    glDisable(GL_TEXTURE_2D);
    glXSwapBuffers(p->display, p->window);
    return RT_OK;
}

void Presenter::poll(Input& in) {
    in.pressed.clear(); in.held.clear(); in.mouseDx = in.mouseDy = 0.0;
    Impl* p = impl_;
    if (!p) { in.quit = true; return; }
    while (XPending(p->display) > 0) {
        XEvent ev;
        XNextEvent(p->display, &ev);
        if (ev.type == KeyPress || ev.type == KeyRelease) {
            if (ev.type == KeyRelease && XPending(p->display) > 0) {
                // auto-repeat arrives as a release immediately followed by a press with the same time stamp: the key stays down
                XEvent next;
                XPeekEvent(p->display, &next);
                if (next.type == KeyPress && next.xkey.time == ev.xkey.time && next.xkey.keycode == ev.xkey.keycode) {
                    XNextEvent(p->display, &next);
                    continue;
                }
            }
            const KeySym sym = XLookupKeysym(&ev.xkey, 0);
            if (sym == XK_Escape) { if (ev.type == KeyPress) in.quit = true; continue; }
            const char c = keyCharacter(sym);
            if (!c) continue;
            const bool press = ev.type == KeyPress;
            if (press && !p->down[(unsigned char)c]) in.pressed.push_back(c);
            p->down[(unsigned char)c] = press;
        } else if (ev.type == MotionNotify) {
            if (p->havePointer) { in.mouseDx += ev.xmotion.x - p->pointerX; in.mouseDy += ev.xmotion.y - p->pointerY; }
            p->havePointer = true;
            p->pointerX = ev.xmotion.x; p->pointerY = ev.xmotion.y;
        } else if (ev.type == ClientMessage) {
            if ((Atom)ev.xclient.data.l[0] == p->deleteMessage) in.quit = true;
        } else if (ev.type == DestroyNotify) {
            in.quit = true;
        }
    }
    for (const char* k = "wasd_|ijkluo"; *k; k++) if (p->down[(unsigned char)*k]) in.held.push_back(*k);
}

void Presenter::setTitle(const std::string& title) {
    if (impl_) XStoreName(impl_->display, impl_->window, title.c_str());
}
