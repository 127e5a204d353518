"""Seeded synthetic frames a fake Testbed returns (inputs of golden g3); shared by
make_goldens.py and the tests so the committed fixture only has to carry the outputs."""
import numpy as np


def seeded_render_frames(seed, h, w):
    """-> (rgba [h,w,4] f32 premultiplied-linear-like, depth [h,w] f32) with the edge cases of
    reference reconstruction/combined_rendering.py:134-153 present: alpha == 0, alpha under
    the 130/255 threshold, depth under 0.05, colours above 1 and slightly negative."""
    r = np.random.Generator(np.random.PCG64(seed))
    a = r.random((h, w), dtype=np.float32)
    a[r.random((h, w)) < 0.15] = 0.0
    a[r.random((h, w)) < 0.15] = 1.0
    rgb = r.random((h, w, 3), dtype=np.float32) * a[..., None] * np.float32(1.1) - np.float32(0.01)
    rgba = np.concatenate([rgb, a[..., None]], -1).astype(np.float32)
    depth = (r.random((h, w), dtype=np.float32) * np.float32(1.5)).astype(np.float32)
    depth[r.random((h, w)) < 0.2] = 0.0
    depth[r.random((h, w)) < 0.1] = np.float32(0.03)
    return rgba, depth
