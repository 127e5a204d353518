"""Physics pre-filter: the oracle's hull-intersection predicate on hand-built
pairs with known answers and its restatement of the reference's control flow (CPU), and the GPU kernel
(wave-per-pose GJK through the C ABI) against that oracle."""
import numpy as np
import pytest

from oracle import host_ref, phys_ref


def box(lo, hi):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def icosphere(centre, r, n=40, seed=0):
    g = np.random.default_rng(seed).standard_normal((n, 3))
    return np.asarray(centre) + r * g / np.linalg.norm(g, axis=1, keepdims=True)


def test_hull_intersection_known_answers():
    unit = box([0, 0, 0], [1, 1, 1])
    assert phys_ref.hulls_intersect(unit, unit + [0.99, 0, 0])                 # overlap 0.01
    assert not phys_ref.hulls_intersect(unit, unit + [1.01, 0, 0])             # gap 0.01
    assert phys_ref.hulls_intersect(unit, unit + [0.9, 0.9, 0.9])              # corner overlap
    assert not phys_ref.hulls_intersect(unit, unit + [1.05, 1.05, 0])          # diagonal neighbours
    # a cube rotated 45 degrees about z has half-diagonal sqrt(2)/2 = 0.7071 along x
    c = (box([-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]) @ rot_z(np.pi / 4).T)
    ax = box([-0.5, -0.5, -0.5], [0.5, 0.5, 0.5])
    assert phys_ref.hulls_intersect(c, ax + [1.2, 0, 0])                        # 0.7071 + 0.5 = 1.2071 > 1.2
    assert not phys_ref.hulls_intersect(c, ax + [1.215, 0, 0])
    # separating axis is an edge-edge cross product (no face normal separates): two long thin boxes, crossed
    a = box([-2, -0.1, -0.1], [2, 0.1, 0.1]) @ rot_z(0.3).T
    b = box([-0.1, -2, 0.12], [0.1, 2, 0.32])
    assert not phys_ref.hulls_intersect(a, b) and phys_ref.hulls_intersect(a, b - [0, 0, 0.05])
    # tetrahedron inside a cube, and a point-like hull
    tet = np.array([[0.4, 0.4, 0.4], [0.6, 0.4, 0.4], [0.4, 0.6, 0.4], [0.4, 0.4, 0.6]])
    assert phys_ref.hulls_intersect(unit, tet) and not phys_ref.hulls_intersect(unit, tet + 2.0)


def test_unique_orientations_follow_the_reference_rule():
    # shelf-type grid: eulers linspace(-pi, pi/2, 3) per axis (obj_pose_opt.py:27-29): (-pi,-pi,-pi) equals the
    # identity (-pi, ..) duplicates show up as equal rotation matrices
    poses = host_ref.sample_poses_grid([0.0, 0.0, 0.0], [1, 1, 1, 3, 3, 3], 1).reshape(-1, 4, 4)
    m = phys_ref.unique_orientation_mask(poses[:, :3, :3])
    R = poses[:, :3, :3]
    assert m[0] and 1 < m.sum() < 27
    for i in np.nonzero(~m)[0]:                       # every dropped one is within 0.01 of an earlier kept one
        assert any(np.abs(R[i] - R[j]).max() <= 0.0101 for j in np.nonzero(m)[0] if j < i)
    kept = np.nonzero(m)[0]
    for a in kept:                                    # kept ones are pairwise distinct
        assert all(np.abs(R[a] - R[b]).max() > 0.0099 for b in kept if b < a)


def _table_scene():
    """A table top (z <= 0), a block standing on it, and a movable box resting on the table at the origin."""
    table = box([-1, -1, -0.1], [1, 1, 0.0])
    block = box([0.30, -0.10, 0.0], [0.50, 0.10, 0.20])
    movable = box([-0.05, -0.05, 0.0], [0.05, 0.05, 0.10]) + [0, 0, 0.005]        # 5 mm above the table at its initial pose
    init = np.eye(4, dtype=np.float32)
    return movable, [table, block], init


def _grid(xs, ys, zs):
    out = []
    for x in xs:
        for y in ys:
            for z in zs:
                T = np.eye(4, dtype=np.float32)
                T[:3, 3] = (x, y, z)
                out.append(T.reshape(16))
    return np.stack(out)


def test_oracle_flow_on_a_table_scene():
    movable, statics, init = _table_scene()
    xs, ys, zs = [0.0, 0.4, 0.95, 1.5], [0.0], [0.0, 0.1, 0.205, -0.5]
    poses = _grid(xs, ys, zs)
    res = [4, 1, 4, 1, 1, 1]
    v = phys_ref.unsupcol_check(poses, init, movable, statics, res, np.ones(16, bool), table_z=-0.3).reshape(4, 4)
    # x = 0: on the table (z offset 0: 5 mm gap -> no collision, lowered 2 cm -> touches, perturbed +-4 cm still over the table) -> valid;
    #        10 cm up: unsupported; 20.5 cm up: unsupported; z = -0.5: pose below table_z -> "supported" unless colliding (it is not: under the slab)
    assert v[0].tolist() == [True, False, False, True]
    # x = 0.4: inside the block at z 0 and 0.1 (collision); at 0.205 it stands on the block: supported, and the +-4 cm probes
    # still overlap the 20 cm wide block -> stable
    assert v[1].tolist() == [False, False, True, True]
    # x = 0.95: on the table's edge: the +x probe (0.99 + 0.05 > 1.0 is still over the table: box spans 0.94..1.04) -> still touching
    assert v[2, 0]
    # x = 1.5: beyond the table: nothing underneath
    assert not v[3, 0] and v[3, 3]
    # a pose that was already invalid stays invalid; stability off accepts edge cases
    v0 = np.ones(16, bool)
    v0[0] = False
    assert not phys_ref.unsupcol_check(poses, init, movable, statics, res, v0, table_z=-0.3)[0]


@pytest.mark.gpu
def test_gpu_prefilter_matches_oracle_on_hulls_and_grids():
    from dream2real_amd import engine, physics_utils
    ctx = engine.Context(0)
    movable, statics, init = _table_scene()
    # (1) pure translations over a dense grid incl. table edge, block top, collisions
    xs = np.linspace(-0.2, 1.2, 15)
    ys = np.linspace(-0.3, 0.3, 5)
    zs = [0.0, 0.012, 0.03, 0.1, 0.2049, 0.23, -0.5]
    poses = _grid(xs, ys, zs)
    res = [15, 5, 7, 1, 1, 1]
    sh = physics_utils.PhysicsShapes(ctx, movable, statics)
    for stab in (True, False):
        got = sh.check(poses, np.ones(len(poses), bool), res, init, -0.3, stability_check=stab)
        want = phys_ref.unsupcol_check(poses, init, movable, statics, res, np.ones(len(poses), bool), -0.3, stability_check=stab)
        assert (got == want).all(), np.nonzero(got != want)[0][:10]
        assert 0.05 < want.mean() < 0.9
    sh.close()
    # (2) six-DoF grid (scene type 1 eulers), rounded hulls, a non-identity initial pose, regrasp rule on and off
    r = np.random.default_rng(5)
    mov = icosphere([0.45, 0.85, 0.27], 0.05, 48, 1) * [1.0, 1.0, 1.6] - [0, 0, 0.16]
    shelf = box([-0.3, 1.1, 0.18], [1.2, 1.5, 0.22])
    blob = icosphere([0.55, 1.27, 0.30], 0.07, 60, 2)
    init = np.eye(4, dtype=np.float32)
    init[:3, :3] = rot_z(0.4).astype(np.float32)
    init[:3, 3] = (0.45, 0.85, 0.2)
    res = [4, 3, 6, 3, 2, 2]
    poses = host_ref.sample_poses_grid([0.45, 0.85, 0.20], res, 1)
    v0 = r.random(len(poses)) > 0.1
    sh = physics_utils.PhysicsShapes(ctx, mov, [shelf, blob])
    for regrasp in (False, True):
        got = sh.check(poses, v0, res, init, 0.2, disallow_regrasp=regrasp)
        want = phys_ref.unsupcol_check(poses, init, mov, [shelf, blob], res, v0, 0.2, disallow_regrasp=regrasp)
        # fp32 GJK vs an LP in double: hulls within ~1e-6 of touching may fall either way
        assert (got != want).mean() < 0.005, ((got != want).sum(), len(want))
        assert (got & ~v0).sum() == 0
    assert want.sum() > 0
    # (3) the closure optimise_pose_grid takes as phys_check
    import types, torch
    task = types.SimpleNamespace(movable_obj=types.SimpleNamespace(pose=torch.tensor(init), phys_hull=mov),
                                 task_bground_obj=types.SimpleNamespace(phys_hulls=[shelf, blob]),
                                 scene_model=types.SimpleNamespace(scene_centre=torch.tensor([0.45, 0.85, 0.20])))
    check, shapes = physics_utils.create_unsupcol_check(ctx, task, res, embodied=False)
    out = check(torch.from_numpy(poses), task, torch.from_numpy(v0))
    assert out.dtype == torch.bool and (out.numpy() == sh.check(poses, v0, res, init, 0.2)).all()
    shapes.close(); sh.close(); ctx.close()
