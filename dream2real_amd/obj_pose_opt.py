"""Candidate-pose grid of the path (reference vision_3d/obj_pose_opt.py:8-55), vectorised numpy.

Poses are absolute world poses.  Order: cartesian product of (x, y, z, rx, ry, rz) with the
last axis fastest; rotation = Rx @ Ry @ Rz (pytorch3d euler_angles_to_matrix(..., 'XYZ'))."""
import math

import numpy as np

# (lo, hi) offsets from scene_centre for x, y, z and absolute euler ranges, per scene type
SCENE_BOUNDS = {
    0: ((-0.12, 0.04), (-0.10, 0.06), (0.00, 0.085), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0)),          # pool table
    1: ((-0.15, 0.20), (0.40, 0.44), (0.04, 0.41),
        (-math.pi, math.pi / 2), (-math.pi, math.pi / 2), (-math.pi, math.pi / 2)),                 # shelf
    3: ((-0.19, 0.15), (-0.25, 0.10), (0.00, 0.14), (0.0, 0.0), (0.0, 0.0), (0.0, 0.0)),           # shopping
}


def linspace_f32(lo, hi, n: int) -> np.ndarray:
    """torch.linspace semantics on float32 (symmetric two-sided fused form)."""
    lo, hi = np.float32(lo), np.float32(hi)
    if n == 1:
        return np.array([lo], np.float32)
    step = np.float64(np.float32((hi - lo) / np.float32(n - 1)))
    i = np.arange(n, dtype=np.float64)
    return np.where(i < n // 2, np.float64(lo) + step * i, np.float64(hi) - step * (n - 1 - i)).astype(np.float32)


def euler_xyz_to_matrix(eulers: np.ndarray) -> np.ndarray:
    """[N,3] -> [N,3,3], Rx(a) @ Ry(b) @ Rz(c), float32."""
    e = np.asarray(eulers, np.float32)
    c, s = np.cos(e), np.sin(e)
    one, zero = np.ones(len(e), np.float32), np.zeros(len(e), np.float32)
    Rx = np.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], -1).reshape(-1, 3, 3)
    Ry = np.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], -1).reshape(-1, 3, 3)
    Rz = np.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], -1).reshape(-1, 3, 3)
    return np.matmul(np.matmul(Rx, Ry), Rz).astype(np.float32)


def sample_poses_grid(task_model, sample_res=(40, 40, 1, 1, 1, 1), scene_type=0) -> np.ndarray:
    """-> [N,16] float32, row-major flattened 4x4 homogeneous poses."""
    if scene_type not in SCENE_BOUNDS:
        raise NotImplementedError("scene_type %d not implemented" % scene_type)
    centre = np.asarray(task_model.scene_model.scene_centre, np.float32).reshape(-1)
    b = SCENE_BOUNDS[scene_type]
    axes = [linspace_f32(np.float32(b[d][0]) + centre[d], np.float32(b[d][1]) + centre[d], int(sample_res[d]))
            for d in range(3)]
    axes += [linspace_f32(b[d][0], b[d][1], int(sample_res[d])) for d in range(3, 6)]
    mesh = np.meshgrid(*axes, indexing="ij")
    combos = np.stack([m.reshape(-1) for m in mesh], axis=-1)          # torch.cartesian_prod order
    poses = np.tile(np.eye(4, dtype=np.float32), (combos.shape[0], 1, 1))
    poses[:, :3, 3] = combos[:, :3]
    poses[:, :3, :3] = euler_xyz_to_matrix(combos[:, 3:])
    return poses.reshape(-1, 16)
