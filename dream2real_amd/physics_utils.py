"""Batched physics pre-filter of candidate poses on the GPU: the counterpart of the reference's
`create_unsupcol_check` / `unsupcol_check` (vision_3d/physics_utils.py:232-375), which walks the N sampled
poses in a Python loop with four to six PyBullet collision queries each.

Same contract: `create_unsupcol_check(...)` returns a closure
`unsupcol_check(pose_batch, task_model, valid_so_far, disallow_regrasp=embodied) -> bool tensor [N]` that
`optimise_pose_grid` takes as `phys_check` (reference dream2real.py:304-326, clip_scoring.py:108-113).
Shapes are convex hulls given as vertex arrays (PyBullet's GEOM_MESH without the concave flag is the convex
hull of the mesh file, :239): `task_model.movable_obj.phys_hull` ([V,3], world frame at the object's initial
pose) and `task_model.task_bground_obj.phys_hulls` (list of [V,3]), or the explicit arguments.  The mesh
pipeline in front of it (TSDF / Poisson / VHACD, :25-229) is outside the path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

GRAVITY_DIRECTION = np.array([0, 0, -1])          # reference vision_3d/physics_utils.py:18


class PhysicsShapes:
    """d2r_phys: the movable hull and the static hulls on the GPU."""

    def __init__(self, ctx, movable_hull, static_hulls):
        self.ctx = ctx
        mov = np.ascontiguousarray(np.asarray(movable_hull, np.float64).reshape(-1, 3), np.float32)
        stat = [np.asarray(h, np.float64).reshape(-1, 3) for h in static_hulls]
        off = np.zeros(len(stat) + 1, np.uint32)
        off[1:] = np.cumsum([len(h) for h in stat])
        sv = np.ascontiguousarray(np.concatenate(stat) if stat else np.zeros((0, 3)), np.float32)
        h = C.c_void_p()
        ctx.check(ctx.lib.d2r_phys_create(ctx.h, _lib.ptr(mov), C.c_uint32(len(mov)), _lib.ptr(sv) if len(sv) else None,
                                          _lib.ptr(off), C.c_uint32(len(stat)), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.d2r_phys_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, pose_batch, valid_so_far, sample_res, init_pose, table_z, unsup_thresh=0.02,
              stability_check=True, disallow_regrasp=False, perturb=0.04) -> np.ndarray:
        poses = np.ascontiguousarray(np.asarray(pose_batch, np.float64).reshape(-1, 16), np.float32)
        valid = np.ascontiguousarray(np.asarray(valid_so_far).astype(np.uint8).reshape(-1))
        assert valid.shape[0] == poses.shape[0]
        prm = _lib.PhysParams((C.c_uint32 * 6)(*[int(x) for x in sample_res]),
                              (C.c_float * 16)(*np.asarray(init_pose, np.float64).reshape(16)),
                              float(table_z), float(unsup_thresh), (C.c_float * 3)(*[float(x) for x in GRAVITY_DIRECTION]),
                              float(perturb), int(bool(stability_check)), int(bool(disallow_regrasp)))
        self.ctx.check(self.ctx.lib.d2r_phys_check(self.ctx.h, self.h, C.byref(prm), _lib.ptr(poses),
                                                   C.c_uint32(poses.shape[0]), _lib.ptr(valid)))
        return valid.astype(bool)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def create_unsupcol_check(ctx, task_model, sample_res, embodied, unsup_thresh=0.02, stability_check=True,
                          movable_hull=None, static_hulls=None):
    """-> (unsupcol_check, shapes).  `ctx` (an engine.Context) takes the place of the reference's
    `pyb_planner`; `lazy_phys_mods` has no counterpart (the hulls given are the ones checked)."""
    if movable_hull is None:
        movable_hull = task_model.movable_obj.phys_hull
    if static_hulls is None:
        static_hulls = task_model.task_bground_obj.phys_hulls
    shapes = PhysicsShapes(ctx, movable_hull, static_hulls)

    def unsupcol_check(pose_batch, task_model, valid_so_far, disallow_regrasp=embodied):
        import torch
        valid = shapes.check(_np(pose_batch), _np(valid_so_far), sample_res, _np(task_model.movable_obj.pose),
                             float(_np(task_model.scene_model.scene_centre)[2]), unsup_thresh, stability_check,
                             disallow_regrasp)
        return torch.from_numpy(valid)

    return unsupcol_check, shapes
