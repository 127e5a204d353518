"""Plain-loop CPU restatements of the host-side steps of the path (small cases only).

TEST INFRASTRUCTURE ONLY — see oracle/d2r_oracle.c.  Each function cites the reference
lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import numpy as np


def converter(T_list):
    """utils/accio2ngp.py:133-139 — flip the y and z camera axes on a copy."""
    out = np.array(T_list, copy=True)
    for T in out:
        for r in range(3):
            T[r, 1] = -T[r, 1]
            T[r, 2] = -T[r, 2]
    return out


def convert_virtual_pose(T_WO_1, T_WO_2, T_WC_1):
    """reconstruction/combined_rendering.py:250-263.  dtype-preserving like numpy there: the
    caller hands T_WO_1 as a float32 [1,4,4] array (:82, from a torch float32 pose), so
    inv(T_WO_1) is a float32 LAPACK inverse while the products run in float64."""
    T_O2_O1 = np.linalg.inv(T_WO_2) @ T_WO_1
    T_O1_C1 = np.linalg.inv(T_WO_1) @ T_WC_1
    return T_WO_1 @ T_O2_O1 @ T_O1_C1


_BOUNDS = {  # vision_3d/obj_pose_opt.py:16-36  (lo, hi) offsets from scene_centre, euler ranges
    0: ((-0.12, 0.04), (-0.10, 0.06), (0.00, 0.085), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0)),
    1: ((-0.15, 0.20), (0.40, 0.44), (0.04, 0.41),
        (-math.pi, math.pi / 2), (-math.pi, math.pi / 2), (-math.pi, math.pi / 2)),
    3: ((-0.19, 0.15), (-0.25, 0.10), (0.00, 0.14), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0)),
}


def _linspace_f32(lo, hi, n):
    """torch.linspace on float32 (ATen CPU RangeFactoriesKernel): step = (hi-lo)/(n-1) in
    float32; first half fma(step, i, lo), second half fma(-step, n-1-i, hi).  The fused
    multiply-add is emulated through float64 (the product of two float32 is exact there);
    checked against torch.linspace in tests/test_host_logic.py."""
    lo, hi = np.float32(lo), np.float32(hi)
    if n == 1:
        return np.array([lo], np.float32)
    step = np.float32((hi - lo) / np.float32(n - 1))
    out = np.zeros(n, np.float32)
    for i in range(n):
        if i < n // 2:
            out[i] = np.float32(np.float64(lo) + np.float64(step) * i)
        else:
            out[i] = np.float32(np.float64(hi) - np.float64(step) * (n - 1 - i))
    return out


def _rot(axis, a):
    c, s = np.float32(math.cos(a)), np.float32(math.sin(a))
    if axis == "X":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float32)
    if axis == "Y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)


def sample_poses_grid(scene_centre, sample_res, scene_type):
    """vision_3d/obj_pose_opt.py:8-55 — linspace per axis, cartesian product in the order
    (x, y, z, rx, ry, rz) with the LAST axis fastest, euler 'XYZ' = Rx @ Ry @ Rz."""
    if scene_type not in _BOUNDS:
        raise NotImplementedError("scene_type %d not implemented" % scene_type)
    b = _BOUNDS[scene_type]
    sc = np.asarray(scene_centre, np.float32)
    axes = []
    for d in range(3):
        lo = np.float32(b[d][0]) + sc[d]
        hi = np.float32(b[d][1]) + sc[d]
        axes.append(_linspace_f32(lo, hi, sample_res[d]))
    for d in range(3, 6):
        axes.append(_linspace_f32(np.float32(b[d][0]), np.float32(b[d][1]), sample_res[d]))
    n = int(np.prod(sample_res))
    out = np.zeros((n, 16), np.float32)
    i = 0
    for x in axes[0]:
        for y in axes[1]:
            for z in axes[2]:
                for rx in axes[3]:
                    for ry in axes[4]:
                        for rz in axes[5]:
                            T = np.eye(4, dtype=np.float32)
                            T[:3, :3] = _rot("X", rx) @ _rot("Y", ry) @ _rot("Z", rz)
                            T[:3, 3] = (x, y, z)
                            out[i] = T.reshape(16)
                            i += 1
    return out


def gaussian_kernel_1d(sigma=0.7):
    """torchvision _get_gaussian_kernel1d(3, sigma): pdf on x in {-1,0,1}, normalised."""
    x = np.array([-1.0, 0.0, 1.0], np.float32)
    pdf = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum()).astype(np.float32)


def spatially_smooth_heatmap(pose_scores, sample_res, sigma=0.7):
    """vision_3d/geometry_utils.py:252-269, explicit loops.

    zeros -> min non-zero; view as [Z*O slices][X][Y]; pad 1 ring with min non-zero; 3x3
    Gaussian (the blur's own reflect padding only touches the ring that is cropped);
    restore the order; re-zero the invalid entries."""
    s = np.array(pose_scores, np.float32, copy=True)
    X, Y = sample_res[0], sample_res[1]
    R = int(np.prod(sample_res[2:]))
    nz = s[s != 0]
    mn = np.float32(nz.min())
    zero = s == 0
    s[zero] = mn
    k1 = gaussian_kernel_1d(sigma)
    k2 = np.outer(k1, k1).astype(np.float32)
    img = s.reshape(X * Y, R)
    out = np.zeros_like(img)
    for r in range(R):
        plane = np.full((X + 2, Y + 2), mn, np.float32)
        plane[1:-1, 1:-1] = img[:, r].reshape(X, Y)
        for i in range(X):
            for j in range(Y):
                acc = np.float32(0.0)
                for di in range(3):
                    for dj in range(3):
                        acc = np.float32(acc + k2[di, dj] * plane[i + di, j + dj])
                out[i * Y + j, r] = acc
    res = out.reshape(-1)
    res[zero] = 0
    return res


def score_logits(all_logits, has_norm):
    """clip_scoring.py:196-203 (no-template branch): goal / mean(norm), or squeeze."""
    a = np.asarray(all_logits, np.float32)
    if not has_norm:
        return a[:, 0].copy()
    out = np.zeros(a.shape[0], np.float32)
    for i in range(a.shape[0]):
        out[i] = a[i, 0] / np.float32(np.mean(a[i, 1:], dtype=np.float32))
    return out


def score_logits_templates(all_logits, n_templates, has_norm):
    """clip_scoring.py:187-195 (template branch)."""
    a = np.asarray(all_logits, np.float32)
    if not has_norm:
        return a.mean(axis=1, dtype=np.float32)
    return a[:, :n_templates].mean(axis=1, dtype=np.float32) / a[:, n_templates:].mean(axis=1, dtype=np.float32)


# ---- cv2.resize(..., interpolation=cv2.INTER_CUBIC) as published (OpenCV imgproc/resize.cpp), scalar ----
# Independent of the product's vectorised dream2real_amd.combined_rendering.resize_cubic: per-pixel loops,
# tap table built the way OpenCV's resize() builds xofs/alpha, float and 8-bit fixed-point paths.
# cv2 is not installed in this image: pinned by hand-derived vectors (tests/test_rectify.py), not by cv2.

def _cv_cubic_coeffs(x):
    """interpolateCubic(): Keys kernel, A = -0.75, float32 arithmetic in OpenCV's expression order
    (numpy float32 scalars: every intermediate is rounded to float32)."""
    f = np.float32
    A, x, one = f(-0.75), f(x), f(1.0)
    x1 = x + one
    c0 = ((A * x1 - f(5) * A) * x1 + f(8) * A) * x1 - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + one
    xm = one - x
    c2 = ((A + f(2)) * xm - (A + f(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return [f(c0), f(c1), f(c2), f(c3)]


def _cv_axis_table(src, dst):
    """resize(): for every destination index the 4 source taps (border-replicated) and coefficients.
    scale = 1 / (dst / src) in double (inv_scale first, as resize() derives it from dsize);
    fx = (float)((d + 0.5) * scale - 0.5); s = floor(fx); fx -= s."""
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    taps, coeffs = [], []
    for d in range(dst):
        fx = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(s))
        taps.append([min(max(s - 1 + k, 0), src - 1) for k in range(4)])
        coeffs.append(_cv_cubic_coeffs(fx))
    return taps, coeffs


def resize_cubic_ref(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_CUBIC) for a 2-D float32 or uint8 image.
    float32: HResizeCubic then VResizeCubic in float (left-to-right sums).
    uint8:   coefficients -> saturate_cast<short>(c * 2048) (round half to even), integer horizontal and
             vertical sums, FixedPtCast<int, uchar, 22>: (v + 2^21) >> 22 saturated to [0, 255]."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape
    xt, xc = _cv_axis_table(sw, dw)
    yt, yc = _cv_axis_table(sh, dh)
    if img.dtype == np.uint8:
        xi = [[int(min(max(np.rint(np.float32(c) * np.float32(2048.0)), -32768), 32767)) for c in cs] for cs in xc]
        yi = [[int(min(max(np.rint(np.float32(c) * np.float32(2048.0)), -32768), 32767)) for c in cs] for cs in yc]
        rows = [[sum(int(img[y, xt[d][k]]) * xi[d][k] for k in range(4)) for d in range(dw)] for y in range(sh)]
        out = np.zeros((dh, dw), np.uint8)
        for e in range(dh):
            for d in range(dw):
                v = sum(rows[yt[e][k]][d] * yi[e][k] for k in range(4))
                out[e, d] = min(max((v + (1 << 21)) >> 22, 0), 255)
        return out
    src = img.astype(np.float32)
    rows = np.zeros((sh, dw), np.float32)
    for y in range(sh):
        for d in range(dw):
            acc = np.float32(src[y, xt[d][0]] * xc[d][0])
            for k in range(1, 4):
                acc = np.float32(acc + np.float32(src[y, xt[d][k]] * xc[d][k]))
            rows[y, d] = acc
    out = np.zeros((dh, dw), np.float32)
    for e in range(dh):
        for d in range(dw):
            acc = np.float32(rows[yt[e][0], d] * yc[e][0])
            for k in range(1, 4):
                acc = np.float32(acc + np.float32(rows[yt[e][k], d] * yc[e][k]))
            out[e, d] = acc
    return out


def rectify_depth_ref(depth, resolution):
    """reconstruction/combined_rendering.py:166-187: centre crop to a square, cubic resize (channel 0 of
    the 4-channel repeat is all the caller reads)."""
    d = np.asarray(depth).astype(np.float32)
    h, w = d.shape
    d = d[(h - w) // 2:(h - w) // 2 + w, :] if h > w else d[:, (w - h) // 2:(w - h) // 2 + h]
    return resize_cubic_ref(np.ascontiguousarray(d), resolution)


def rectify_mask_ref(mask, resolution):
    """reconstruction/combined_rendering.py:189-209: same crop + cubic resize on the uint8 mask."""
    m = np.asarray(mask).astype(np.uint8)
    h, w = m.shape
    m = m[(h - w) // 2:(h - w) // 2 + w, :] if h > w else m[:, (w - h) // 2:(w - h) // 2 + h]
    return resize_cubic_ref(np.ascontiguousarray(m), resolution)
