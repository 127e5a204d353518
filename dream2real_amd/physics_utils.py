"""Batched physics pre-filter of candidate poses on the GPU: the counterpart of the reference's
`create_unsupcol_check` / `unsupcol_check` (vision_3d/physics_utils.py:232-375), which walks the N sampled
poses in a Python loop with four to six PyBullet collision queries each.

Same contract: `create_unsupcol_check(ctx, task_model, sample_res, embodied, unsup_thresh, lazy_phys_mods,
stability_check)` returns `(unsupcol_check, static_obj_handles, movable_handles)` with
`unsupcol_check(pose_batch, task_model, valid_so_far, disallow_regrasp=embodied) -> bool tensor [N]`, the closure
`optimise_pose_grid` takes as `phys_check` (reference dream2real.py:304-326, clip_scoring.py:108-113).  `ctx` (an
engine.Context) stands where the reference passes `pyb_planner`.

Shapes come from where the reference takes them: every object's `phys_model` is the path of a Wavefront .obj mesh in
world coordinates (reference :238 `createCollisionShape(GEOM_MESH, fileName=obj.phys_model)`; written by
get_phys_models :25-229 — TSDF / Poisson / VHACD, outside the path).  PyBullet turns each shape of the file (an `o` /
`g` group: VHACD writes one per convex part) into the convex hull of its vertices, so an object is a compound of
convex parts: `hulls_from_obj` reads exactly that.  Objects may instead carry vertex arrays (`phys_hull` /
`phys_hulls`), which take precedence — tests and callers without mesh files use them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

GRAVITY_DIRECTION = np.array([0, 0, -1])          # reference vision_3d/physics_utils.py:18
# Collision margin PyBullet gives a convex hull it loads from a mesh file (believed: its default collision margin for
# file-loaded convex shapes is 0.001 m, and getClosestPoints at distance 0 — what pairwise_collision asks — reports
# two bodies whose hulls are closer than the sum of their margins).  PyBullet is not available offline: UNPINNED.
PYBULLET_MESH_MARGIN = 0.001


def hulls_from_obj(path: str) -> list:
    """Vertex sets of the convex parts of a Wavefront .obj, one per shape as PyBullet's GEOM_MESH loader (tinyobj)
    splits the file: a new shape starts at every `o` or `g` line, and a shape's vertices are the ones its faces
    reference (indices are global, 1-based, negative = relative to the vertices read so far; `v/vt/vn` forms
    accepted).  A file without groups is one shape; a shape without faces is skipped; a file with vertices but no
    faces at all is one shape of all its vertices (a point cloud's hull)."""
    verts, shapes, cur = [], [], set()

    def close():
        if cur:
            shapes.append(sorted(cur))
        cur.clear()

    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v" and len(t) >= 4:
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] in ("o", "g"):
                close()
            elif t[0] == "f":
                for tok in t[1:]:
                    i = int(tok.split("/")[0])
                    i = i - 1 if i > 0 else len(verts) + i
                    if not 0 <= i < len(verts):
                        raise ValueError(f"{path}: face references vertex {tok} of {len(verts)}")
                    cur.add(i)
    close()
    v = np.asarray(verts, np.float64).reshape(-1, 3)
    if not shapes:
        if len(v) == 0:
            raise ValueError(f"{path}: no vertices")
        return [v]
    return [v[idx] for idx in shapes]


def object_hulls(obj) -> list:
    """Convex parts of a scene object: `phys_hulls` (list of [V,3]) or `phys_hull` ([V,3]) when the object carries
    vertex arrays, else the shapes of its `phys_model` mesh file (reference scene_model.py:14-17)."""
    if getattr(obj, "phys_hulls", None) is not None:
        return [np.asarray(h, np.float64).reshape(-1, 3) for h in obj.phys_hulls]
    if getattr(obj, "phys_hull", None) is not None:
        return [np.asarray(obj.phys_hull, np.float64).reshape(-1, 3)]
    model = getattr(obj, "phys_model", None)
    if model is None:
        raise ValueError("object has neither hull vertex arrays nor a phys_model mesh path")
    return hulls_from_obj(model)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _as_parts(hulls, what):
    """One hull ([V,3] array, nested list or tensor) or a sequence of hulls -> list of [V,3] float64 arrays.  A bare
    [V,3] list-of-lists is ONE hull (iterating it would give V one-point 'hulls' that pass validation and collide wrongly)."""
    if hasattr(hulls, "detach") or isinstance(hulls, np.ndarray):
        hulls = _np(hulls)
    else:
        try:
            arr = np.asarray([_np(h) for h in hulls], np.float64)
        except (ValueError, TypeError):
            arr = None                                   # ragged: a real sequence of hulls
        if arr is not None and arr.ndim == 2 and arr.shape[1] == 3:
            hulls = arr
    if isinstance(hulls, np.ndarray):
        if hulls.ndim == 2 and hulls.shape[1] == 3:
            hulls = [hulls]
        elif hulls.ndim == 3 and hulls.shape[2] == 3:
            hulls = list(hulls)
        else:
            raise ValueError(f"{what}: expected [V,3] vertices or a sequence of such arrays, got shape {hulls.shape}")
    parts = []
    for k, h in enumerate(hulls):
        a = np.asarray(_np(h), np.float64)
        if a.ndim != 2 or a.shape[1] != 3 or a.shape[0] < 1:
            raise ValueError(f"{what}: part {k} must be a [V,3] array with at least one vertex, got shape {a.shape}")
        parts.append(a)
    return parts


def _pack(hulls):
    hs = [np.asarray(h, np.float64).reshape(-1, 3) for h in hulls]
    off = np.zeros(len(hs) + 1, np.uint32)
    off[1:] = np.cumsum([len(h) for h in hs])
    v = np.ascontiguousarray(np.concatenate(hs) if hs else np.zeros((0, 3)), np.float32)
    return v, off


class PhysicsShapes:
    """d2r_phys: the movable object's convex part(s) and the static parts on the GPU."""

    def __init__(self, ctx, movable_hulls, static_hulls):
        self.ctx = ctx
        movable_hulls = _as_parts(movable_hulls, "movable_hulls")      # a single hull in any array-like form, or parts
        static_hulls = _as_parts(static_hulls, "static_hulls") if len(static_hulls) else []
        mv, moff = _pack(movable_hulls)
        sv, soff = _pack(static_hulls)
        h = C.c_void_p()
        ctx.check(ctx.lib.d2r_phys_create(ctx.h, _lib.ptr(mv), _lib.ptr(moff), C.c_uint32(len(moff) - 1),
                                          _lib.ptr(sv) if len(sv) else None, _lib.ptr(soff), C.c_uint32(len(soff) - 1), C.byref(h)))
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
              stability_check=True, disallow_regrasp=False, perturb=0.04, margin=0.0) -> np.ndarray:
        poses = np.ascontiguousarray(np.asarray(pose_batch, np.float64).reshape(-1, 16), np.float32)
        valid = np.ascontiguousarray(np.asarray(valid_so_far).astype(np.uint8).reshape(-1))
        assert valid.shape[0] == poses.shape[0]
        prm = _lib.PhysParams((C.c_uint32 * 6)(*[int(x) for x in sample_res]),
                              (C.c_float * 16)(*np.asarray(init_pose, np.float64).reshape(16)),
                              float(table_z), float(unsup_thresh), (C.c_float * 3)(*[float(x) for x in GRAVITY_DIRECTION]),
                              float(perturb), int(bool(stability_check)), int(bool(disallow_regrasp)), float(margin))
        self.ctx.check(self.ctx.lib.d2r_phys_check(self.ctx.h, self.h, C.byref(prm), _lib.ptr(poses),
                                                   C.c_uint32(poses.shape[0]), _lib.ptr(valid)))
        return valid.astype(bool)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def create_unsupcol_check(ctx, task_model, sample_res, embodied, unsup_thresh=0.02, lazy_phys_mods=True, stability_check=True,
                          margin=PYBULLET_MESH_MARGIN, movable_hull=None, static_hulls=None):
    """-> (unsupcol_check, static_obj_handles, movable_handles), the reference's triple (vision_3d/physics_utils.py:232,377).

    Objects checked, as the reference chooses them (:235): with `lazy_phys_mods` the merged background object and the
    movable object, otherwise every object of the scene model — all but the movable one static.  The handles are what
    stands for PyBullet's body ids here: per static object the list of its convex parts' vertex arrays, and for the
    movable object a one-element list holding its parts; `unsupcol_check.shapes` is the GPU-side object (close() frees it).
    `movable_hull` / `static_hulls` override the lookup with explicit vertex arrays."""
    if movable_hull is not None or static_hulls is not None:
        mov = [np.asarray(movable_hull, np.float64).reshape(-1, 3)] if movable_hull is not None else object_hulls(task_model.movable_obj)
        static_objs = [[np.asarray(h, np.float64).reshape(-1, 3)] for h in static_hulls] if static_hulls is not None else \
            [object_hulls(task_model.task_bground_obj)]
    else:
        phys_obj_list = [task_model.task_bground_obj, task_model.movable_obj] if lazy_phys_mods else list(task_model.scene_model.objs)
        mov, static_objs = None, []
        for obj in phys_obj_list:
            if obj is task_model.movable_obj:
                mov = object_hulls(obj)
            else:
                static_objs.append(object_hulls(obj))
        if mov is None:
            raise ValueError("the movable object is not among the objects to check")
    shapes = PhysicsShapes(ctx, mov, [h for parts in static_objs for h in parts])

    def unsupcol_check(pose_batch, task_model, valid_so_far, disallow_regrasp=embodied):
        import torch
        valid = shapes.check(_np(pose_batch), _np(valid_so_far), sample_res, _np(task_model.movable_obj.pose),
                             float(_np(task_model.scene_model.scene_centre)[2]), unsup_thresh, stability_check,
                             disallow_regrasp, margin=margin)
        return torch.from_numpy(valid)

    unsupcol_check.shapes = shapes
    return unsupcol_check, static_objs, [mov]
