"""Heat-map smoothing of the path (reference vision_3d/geometry_utils.py:252-269)."""
import numpy as np


def gaussian_kernel_3(sigma: float = 0.7) -> np.ndarray:
    """torchvision gaussian_blur(kernel_size=3): normalised pdf at x in {-1,0,1}, outer product."""
    x = np.array([-1.0, 0.0, 1.0], np.float32)
    k = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    k = (k / k.sum()).astype(np.float32)
    return np.outer(k, k).astype(np.float32)


def spatially_smooth_heatmap(pose_scores, sample_res, sigma: float = 0.7) -> np.ndarray:
    """3x3 Gaussian over the (x, y) plane of every (z, orientation) slice.  Invalid (zero)
    scores and the 1-cell border take the minimum non-zero score; invalid cells are zeroed
    again afterwards.  pose_scores: [prod(sample_res)] -> same shape, float32."""
    s = np.array(pose_scores, np.float32, copy=True).reshape(-1)
    X, Y = int(sample_res[0]), int(sample_res[1])
    R = int(np.prod([int(v) for v in sample_res[2:]]))
    zero = s == 0
    mn = s[~zero].min()
    s[zero] = mn
    planes = s.reshape(X, Y, R)
    padded = np.full((X + 2, Y + 2, R), mn, np.float32)
    padded[1:-1, 1:-1] = planes
    k = gaussian_kernel_3(sigma)
    out = np.zeros((X, Y, R), np.float32)
    for di in range(3):
        for dj in range(3):
            out += k[di, dj] * padded[di:di + X, dj:dj + Y]
    out = out.reshape(-1)
    out[zero] = 0
    return out
