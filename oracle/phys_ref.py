"""CPU restatement of the reference's physics pre-filter control flow, `unsupcol_check`
(vision_3d/physics_utils.py:248-375), with convex-hull intersection as the collision predicate.

TEST INFRASTRUCTURE ONLY — see oracle/d2r_oracle.c.  The reference asks PyBullet
(`pyb_planner.pairwise_collision`, :318,:338,:357) whether two GEOM_MESH bodies — compounds of the convex hulls
of their mesh files' shapes (:239, no GEOM_FORCE_CONCAVE_TRIMESH) — are in contact.  PyBullet is not installed
here: PARITY UNPINNED against PyBullet; its collision margin is a parameter (`margin`, per convex part: contact
when two hulls are closer than 2 * margin; what value PyBullet uses for file-loaded hulls is a belief, see
dream2real_amd/physics_utils.PYBULLET_MESH_MARGIN).  The predicates are pinned by hand-built hull pairs with known
answers (tests/test_physics.py).  Deliberately independent of the GPU kernel's GJK: intersection is decided as a
linear-programming feasibility problem (is there a point that is a convex combination of both vertex sets?),
distance as a small quadratic program.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linprog

GRAVITY_DIRECTION = np.array([0, 0, -1])        # vision_3d/physics_utils.py:18


def hull_distance(a: np.ndarray, b: np.ndarray) -> float:
    """Euclidean distance between conv(a) and conv(b): min |a^T l - b^T m| over the two simplices, as a small QP
    (SLSQP from several starts; the problem is convex, the starts guard against a stalled line search)."""
    from scipy.optimize import minimize
    a = np.asarray(a, np.float64).reshape(-1, 3)
    b = np.asarray(b, np.float64).reshape(-1, 3)
    na, nb = len(a), len(b)
    M = np.concatenate([a, -b]).T                                  # 3 x (na + nb)
    cons = [{"type": "eq", "fun": lambda x: x[:na].sum() - 1.0, "jac": lambda x: np.r_[np.ones(na), np.zeros(nb)]},
            {"type": "eq", "fun": lambda x: x[na:].sum() - 1.0, "jac": lambda x: np.r_[np.zeros(na), np.ones(nb)]}]
    best = np.inf
    ca, cb = a.mean(0), b.mean(0)
    starts = [np.r_[np.full(na, 1.0 / na), np.full(nb, 1.0 / nb)]]
    x0 = np.zeros(na + nb)
    x0[np.argmax(a @ (cb - ca))] = 1.0                             # the vertices facing the other hull
    x0[na + np.argmax(b @ (ca - cb))] = 1.0
    starts.append(x0)
    for x0 in starts:
        r = minimize(lambda x: 0.5 * float((M @ x) @ (M @ x)), x0, jac=lambda x: M.T @ (M @ x), bounds=[(0, 1)] * (na + nb),
                     constraints=cons, method="SLSQP", options={"maxiter": 500, "ftol": 1e-16})
        best = min(best, float(np.linalg.norm(M @ np.clip(r.x, 0, 1))))
    return best


def hulls_intersect(a: np.ndarray, b: np.ndarray, margin: float = 0.0) -> bool:
    """conv(a) and conv(b) share a point  <=>  exist l, m >= 0, sum l = sum m = 1, a^T l = b^T m (an LP).
    margin > 0 (each shape's collision margin): in contact when the hulls are closer than 2 * margin."""
    a = np.asarray(a, np.float64).reshape(-1, 3)
    b = np.asarray(b, np.float64).reshape(-1, 3)
    if margin > 0.0:
        lo_a, hi_a, lo_b, hi_b = a.min(0), a.max(0), b.min(0), b.max(0)
        if (lo_a > hi_b + 2 * margin).any() or (lo_b > hi_a + 2 * margin).any():
            return False
        return hull_distance(a, b) <= 2.0 * margin
    lo_a, hi_a, lo_b, hi_b = a.min(0), a.max(0), b.min(0), b.max(0)
    if (lo_a > hi_b).any() or (lo_b > hi_a).any():
        return False
    na, nb = len(a), len(b)
    A_eq = np.zeros((5, na + nb))
    A_eq[:3, :na] = a.T
    A_eq[:3, na:] = -b.T
    A_eq[3, :na] = 1.0
    A_eq[4, na:] = 1.0
    b_eq = np.array([0, 0, 0, 1, 1], np.float64)
    res = linprog(np.zeros(na + nb), A_eq=A_eq, b_eq=b_eq, bounds=(0, None), method="highs")
    return res.status == 0


def unique_orientation_mask(first_pos_oris: np.ndarray) -> np.ndarray:
    """:260-278 — greedy: an orientation is dropped when torch.isclose(ori, seen, atol=0.01) holds for all nine
    entries against an orientation kept earlier (isclose: |a - b| <= atol + rtol |b|, rtol 1e-5)."""
    kept, mask = [], np.ones(len(first_pos_oris), bool)
    for i, ori in enumerate(first_pos_oris):
        seen = False
        for s in kept:
            if np.all(np.abs(ori - s) <= 0.01 + 1e-5 * np.abs(s)):
                seen = True
                break
        if seen:
            mask[i] = False
        else:
            kept.append(ori)
    return mask


def unsupcol_check(pose_batch, init_pose, movable_hull, static_hulls, sample_res, valid_so_far, table_z,
                   disallow_regrasp=False, unsup_thresh=0.02, stability_check=True, margin=0.0):
    """vision_3d/physics_utils.py:248-375, line by line; returns the bool mask [N].  movable_hull: one vertex array, or
    a list of them (a compound of convex parts: two bodies touch when any pair of parts does)."""
    valid = np.array(valid_so_far, bool, copy=True)
    poses = np.asarray(pose_batch, np.float32).reshape(-1, 4, 4)
    transforms = poses.astype(np.float64) @ np.linalg.inv(np.asarray(init_pose, np.float64).reshape(4, 4))   # :253
    n_pos = sample_res[0] * sample_res[1] * sample_res[2]
    n_ori = sample_res[3] * sample_res[4] * sample_res[5]
    first = poses[:n_ori, :3, :3]
    m1 = unique_orientation_mask(first)                                  # :260-276
    valid &= np.tile(m1, n_pos)                                          # :277-278
    m2 = np.ones(n_ori, bool)
    if disallow_regrasp:                                                 # :283-297
        for i in range(n_ori):
            if not valid[i]:
                m2[i] = False
                continue
            z = first[i][:, 2]
            facing = (z @ np.array([0, 0, 1.0], np.float32) > 0.9) or (z @ np.array([0, -1.0, 0], np.float32) > 0.9)
            if not facing:
                m2[i] = False
    valid &= np.tile(m2, n_pos)                                          # :300-301
    movs = [np.asarray(m, np.float64).reshape(-1, 3) for m in movable_hull] if isinstance(movable_hull, (list, tuple)) else \
        [np.asarray(movable_hull, np.float64).reshape(-1, 3)]

    def touches_any(R, t):
        return any(hulls_intersect(m @ R.T + t, h, margin) for m in movs for h in static_hulls)

    for i in range(len(poses)):                                          # :308
        if not valid[i]:
            continue
        R, pos = transforms[i, :3, :3], transforms[i, :3, 3]
        if touches_any(R, pos):                                          # :314-321 collision
            valid[i] = False
            continue
        lower = pos + unsup_thresh * GRAVITY_DIRECTION                   # :329
        below_table = poses[i, 2, 3] < table_z                           # :332-333
        valid[i] = below_table or touches_any(R, lower)                  # :334-340
        if not valid[i]:
            continue
        if stability_check and not below_table:                          # :349-365
            for pv in (np.array([1, 0, 0]), np.array([-1, 0, 0]), np.array([0, 1, 0]), np.array([0, -1, 0])):
                if not touches_any(R, lower + 0.04 * pv):
                    valid[i] = False
                    break
    return valid
