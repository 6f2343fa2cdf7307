// rt_headless: the reference's main() (main.cpp:11-23) without a window.
//   rt_headless [--size WxH] [--preset KEY] [--frames N] [--models DIR] [--out frame.ppm] [--device N] [--gpus N] [--sharded]
//               [--keys STRING] [--sequence DIR]
// --keys holds one movement key per frame, cyclically (w a s d _ | move the camera, i j k l u o the light, < > turn the
// camera, a blank nothing): the reference's interactive frame loop without the window.  --sequence writes every frame of
// the run into DIR as frame_0000.ppm, ...
// Renders the preset's scene N times (default 100, as the reference's render-time average) and prints the timing.
// --gpus N shards every frame by image tiles over GPUs 0 .. N-1 of this node: one process per GPU (forked before anything
// touches a GPU), each with its own copy of the scene; the tiles are gathered to rank 0 over RCCL and rank 0 reports.
// --sharded takes the same path (tile buffer, gather, reassembly) with a single GPU: a rehearsal of the exchange.
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Headless.h"
#include "rt_rccl.h"

namespace {

bool readAll(int fd, void* buf, size_t n) {
    char* p = static_cast<char*>(buf);
    while (n) { ssize_t r = read(fd, p, n); if (r <= 0) return false; p += r; n -= (size_t)r; }
    return true;
}
bool writeAll(int fd, const void* buf, size_t n) {
    const char* p = static_cast<const char*>(buf);
    while (n) { ssize_t r = write(fd, p, n); if (r <= 0) return false; p += r; n -= (size_t)r; }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    int width = 1000, height = 1000, frames = 100, device = 0, gpus = 1;   // main.cpp:13 opens a 1000x1000 window
    bool sharded = false;
    std::string models = "assets/models", out, keys, sequence;
    char preset = 'Z';
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "--size")) { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (!std::strcmp(argv[i], "--preset")) preset = next()[0];
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next());
        else if (!std::strcmp(argv[i], "--models")) models = next();
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(next());
        else if (!std::strcmp(argv[i], "--sharded")) sharded = true;
        else if (!std::strcmp(argv[i], "--keys")) keys = next();
        else if (!std::strcmp(argv[i], "--sequence")) sequence = next();
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (gpus < 1 || gpus > 64) { std::fprintf(stderr, "--gpus 1 .. 64\n"); return 2; }

    // One process per GPU.  The children are forked here, before any HIP or RCCL call; each gets a pipe on which rank 0 sends
    // the communicator's identifier once it has made it.
    int rank = 0;
    std::vector<int> toChild;            // rank 0: write ends, one per child
    std::vector<pid_t> children;
    int fromParent = -1;
    for (int r = 1; r < gpus && rank == 0; r++) {
        int fds[2];
        if (pipe(fds) != 0) { std::perror("pipe"); return 1; }
        pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) {
            rank = r; fromParent = fds[0]; close(fds[1]);
            for (int fd : toChild) close(fd);
            toChild.clear(); children.clear();
        } else {
            close(fds[0]); toChild.push_back(fds[1]); children.push_back(pid);
        }
    }
    if (gpus > 1) { device = rank; sharded = true; }

    rt_comm* comm = nullptr;
    if (sharded) {
        unsigned char id[RT_COMM_ID_BYTES];
        if (rank == 0) {
            if (rt_comm_make_id(id) != RT_OK) { std::fprintf(stderr, "rank 0: %s\n", rt_comm_last_error()); return 1; }
            for (int fd : toChild) { if (!writeAll(fd, id, sizeof id)) { std::fprintf(stderr, "rank 0: cannot reach a child\n"); return 1; } close(fd); }
        } else if (!readAll(fromParent, id, sizeof id)) {
            std::fprintf(stderr, "rank %d: no communicator identifier from rank 0\n", rank); return 1;
        }
        if (rt_comm_create(id, rank, gpus, device, &comm) != RT_OK) { std::fprintf(stderr, "rank %d: %s\n", rank, rt_comm_last_error()); return 1; }
    }

    int status = 0;
    {
        Headless app(width, height, models, device);
        if (comm) {
            app.rayTracer.renderSharded = [comm](rt_ctx* ctx, int w, int h, GLubyte* fb) {
                const int rc = rt_render_frame_sharded(ctx, comm, w, h, fb);
                if (rc != RT_OK) std::fprintf(stderr, "sharded frame: %s\n", rt_comm_last_error());
                return rc;
            };
        }
        if (!app.keyPreset(preset)) { std::fprintf(stderr, "unknown preset key '%c' (Z X C V B N M , .)\n", preset); return 2; }
        const bool moving = !keys.empty() || !sequence.empty();
        double avg = moving ? app.runSequence(frames, keys, rank == 0 ? sequence : std::string()) : app.measure(frames);
        if (avg < 0.0) { std::fprintf(stderr, "cannot write into %s\n", sequence.c_str()); status = 1; avg = 0.0; }
        if (rank == 0) {
            if (comm) std::printf("Average render time: %f over %d renders on %d GPUs (image tiles, RCCL gather to rank 0)\n", avg, frames, gpus);
            else std::printf("Average render time: %f over %d renders (last frame: %.3f ms on the GPU, %llu rays, %.1f Mrays/s)\n", avg, frames,
                             app.rayTracer.lastGpuMs, app.rayTracer.lastRayCount,
                             app.rayTracer.lastGpuMs > 0 ? app.rayTracer.lastRayCount / (app.rayTracer.lastGpuMs * 1e3) : 0.0);
            if (!out.empty() && !app.writePPM(out)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); status = 1; }
        }
    }
    if (comm) rt_comm_destroy(comm);
    for (pid_t pid : children) { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = 1; }
    return status;
}
