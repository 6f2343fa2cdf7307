"""rt_render_device_static - the entry point that allocates nothing - against rt_render_device, through the C ABI.

Two calls in a row on one context, the second with a smaller frame (the shared device array then still holds the first
call's costs and order), then the first size again twice so that a frame is also dispatched in the order the array holds.
Bar: byte-identical frames (integer output of the same kernels)."""
import numpy as np
import pytest

from conftest import load_package

load_package()
from cuda_ray_tracer_amd import binding, scenes  # noqa: E402

pytestmark = pytest.mark.gpu


def _pair(renderer, w, h):
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda:0")
    b = torch.full((w * h * 3,), 7, dtype=torch.uint8, device="cuda:0")
    renderer.render_device(w, h, a.data_ptr(), stream)
    renderer.render_device_static(w, h, b.data_ptr(), stream)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_scratch_entry_point_matches_render_device(pkg):
    import torch  # noqa: F401  (first, so that the library shares torch's HIP runtime)
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.2)
    built = scenes.build_scene(spec)
    r = binding.Renderer(0)
    try:
        r.upload(built.desc)
        r.set_state(built.camera, cfg)
        for w, h in [(640, 360), (333, 201), (640, 360), (640, 360)]:
            ref, got = _pair(r, w, h)
            assert ref.any(), "the reference frame is empty"
            assert np.array_equal(ref, got), f"{w}x{h}: static-scratch frame differs from rt_render_device's"
        # a second context on the same GPU shares the array
        r2 = binding.Renderer(0)
        try:
            r2.upload(built.desc)
            r2.set_state(built.camera, cfg)
            ref, got = _pair(r2, 640, 360)
            assert np.array_equal(ref, got)
        finally:
            r2.close()
        # beyond the array's capacity: 8192 x 8192 = 65536 tiles
        with pytest.raises(binding.RtError, match="static scratch"):
            r.render_device_static(8192, 8192, 16, None)
    finally:
        r.close()


def test_two_contexts_on_two_streams_share_the_array_safely(pkg):
    """Two contexts, each on a stream of its own, alternate through the entry point with different frame sizes and without a
    synchronise in between: the shared order list must never be read or ranked while the other stream's launches still use
    it (a torn list leaves tiles unrendered, i.e. the buffer's fill value).  Compared with rt_render_device frames."""
    import torch
    spec, cfg, _, _ = scenes.config_scene(3, detail=0.2)
    built = scenes.build_scene(spec)
    sizes = [(640, 360), (512, 288)]
    ctxs = [binding.Renderer(0), binding.Renderer(0)]
    streams = [torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")]
    try:
        want = []
        for r, (w, h) in zip(ctxs, sizes):
            r.upload(built.desc)
            r.set_state(built.camera, cfg)
            buf = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda:0")
            r.render_device(w, h, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want.append(buf.cpu().numpy())
        rounds = 6
        outs = [[torch.full((w * h * 3,), 7, dtype=torch.uint8, device="cuda:0") for _ in range(rounds)] for (w, h) in sizes]
        for k in range(rounds):
            for i, (r, (w, h)) in enumerate(zip(ctxs, sizes)):
                r.render_device_static(w, h, outs[i][k].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        for i in range(2):
            for k in range(rounds):
                assert np.array_equal(outs[i][k].cpu().numpy(), want[i]), f"context {i}, frame {k}: differs from rt_render_device's"
        # one context, frames alternating between a caller's stream and the blocking entry point's own stream
        r = ctxs[0]
        w, h = sizes[0]
        for k in range(4):
            dev = torch.full((w * h * 3,), 7, dtype=torch.uint8, device="cuda:0")
            r.render_device(w, h, dev.data_ptr(), streams[1].cuda_stream)
            host, _ = r.render(w, h)
            torch.cuda.synchronize()
            assert np.array_equal(host.reshape(-1), want[0]) and np.array_equal(dev.cpu().numpy(), want[0])
    finally:
        for r in ctxs:
            r.close()
