"""Properties of the OpenCV-style cubic resize used by rectify_depth / rectify_mask
(reference combined_rendering.py:166-209).  cv2 is not installed, so these are mathematical
properties of INTER_CUBIC, not a pin against OpenCV."""
import numpy as np

from dream2real_amd.combined_rendering import _cubic_taps, rectify_depth, rectify_mask, resize_cubic


def test_identity_and_constants():
    r = np.random.default_rng(0)
    img = r.random((20, 30), dtype=np.float32)
    np.testing.assert_allclose(resize_cubic(img, (30, 20)), img, atol=1e-6)       # same size: fx == 0
    c = np.full((50, 70), 1.25, np.float32)
    np.testing.assert_allclose(resize_cubic(c, (33, 21)), 1.25, atol=1e-5)        # partition of unity
    u = np.full((50, 70), 201, np.uint8)
    assert (resize_cubic(u, (33, 21)) == 201).all()


def test_coefficients_known_values():
    idx, c = _cubic_taps(4, 8)               # scale 0.5: fx = -0.25, 0.25, ...
    np.testing.assert_allclose(c.sum(1), 1.0, atol=1e-6)
    # x = 0.75 (dst 0: fx=-0.25 -> sx=-1, frac 0.75), Keys A=-0.75
    x, A = 0.75, -0.75
    want1 = ((A + 2) * x - (A + 3)) * x * x + 1
    assert abs(c[0, 1] - want1) < 1e-6
    assert idx[0].tolist() == [0, 0, 0, 1]                                         # replicated border
    assert idx[-1].tolist()[-1] == 3


def test_linear_ramp_is_reproduced_in_the_interior():
    ramp = np.tile(np.arange(64, dtype=np.float32), (8, 1))
    out = resize_cubic(ramp, (32, 8))
    x = (np.arange(32) + 0.5) * 2 - 0.5
    np.testing.assert_allclose(out[0, 2:-2], x[2:-2], atol=1e-4)                   # cubic convolution is exact on linear data


def test_rectify_shapes_and_crop():
    depth = np.zeros((720, 1280), np.float16)
    depth[:, 280:1000] = 0.5                                                       # the centred 720x720 square
    d = rectify_depth(depth, (336, 336))
    assert d.shape == (336, 336) and d.dtype == np.float32
    np.testing.assert_allclose(d, 0.5, atol=1e-3)
    mask = np.zeros((720, 1280), bool)
    mask[200:500, 500:800] = True
    m = rectify_mask(mask, (336, 336))
    assert m.dtype == np.uint8 and set(np.unique(m)) <= {0, 1}
    assert m[160, 160] == 1 and m[10, 10] == 0
