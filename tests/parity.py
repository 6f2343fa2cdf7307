"""Shared helpers for GPU-vs-oracle parity checks."""
import numpy as np


def compare_frames(gpu_rgb, gpu_ids, gpu_t, ora_rgb, ora_ids, ora_t, window=None):
    """Returns a dict of mismatch statistics between a GPU frame and an oracle frame (optionally inside a window)."""
    if window:
        x0, y0, x1, y1 = window
        sl = (slice(y0, y1), slice(x0, x1))
        gpu_rgb, ora_rgb = gpu_rgb[sl], ora_rgb[sl]
        if gpu_ids is not None:
            gpu_ids, ora_ids, gpu_t, ora_t = gpu_ids[sl], ora_ids[sl], gpu_t[sl], ora_t[sl]
    diff = np.abs(gpu_rgb.astype(np.int16) - ora_rgb.astype(np.int16))
    px = diff.max(axis=-1)
    out = {
        "pixels": int(px.size),
        "max_abs": int(diff.max()) if diff.size else 0,
        "px_gt0": int((px > 0).sum()),
        "px_gt1": int((px > 1).sum()),
    }
    if gpu_ids is not None:
        same = np.all(gpu_ids == ora_ids, axis=-1)
        out["id_mismatch"] = int((~same).sum())
        tg, to = gpu_t.view(np.uint32), ora_t.view(np.uint32)
        out["t_bit_mismatch"] = int((tg != to).sum())
    return out


def assert_parity(stats, rays_gpu=None, rays_oracle=None, max_px_gt1_frac=0.0, max_id_frac=1e-4):
    """The parity bar: RGB within 1 LSB (8-bit) and primary-hit ids identical on >= 99.99 % of pixels."""
    n = stats["pixels"]
    assert stats["px_gt1"] <= max_px_gt1_frac * n, f"RGB differs by more than 1 LSB on {stats['px_gt1']} of {n} pixels: {stats}"
    if "id_mismatch" in stats:
        assert stats["id_mismatch"] <= max_id_frac * n, f"primary-hit ids differ on {stats['id_mismatch']} of {n} pixels: {stats}"
    if rays_gpu is not None:
        assert rays_gpu == rays_oracle, f"ray count differs: gpu {rays_gpu} vs oracle {rays_oracle}"
