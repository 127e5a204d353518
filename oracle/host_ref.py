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
