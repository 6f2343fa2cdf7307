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
