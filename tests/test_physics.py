"""Physics pre-filter: the oracle's hull-intersection predicate on hand-built
pairs with known answers and its restatement of the reference's control flow (CPU), and the GPU kernel
(wave-per-pose GJK through the C ABI) against that oracle."""
import numpy as np
import pytest

from oracle import host_ref, phys_ref


from synthetic_scenes import box, icosphere  # noqa: E402,F401


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


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


def test_hull_distance_and_margins_known_answers():
    """Contact with collision margins: two parts touch when their hulls are closer than the sum of the margins."""
    unit = box([0, 0, 0], [1, 1, 1])
    assert abs(phys_ref.hull_distance(unit, unit + [1.25, 0, 0]) - 0.25) < 1e-7            # face to face
    assert abs(phys_ref.hull_distance(unit, unit + [1.3, 1.4, 0]) - 0.5) < 1e-7             # edge to edge: (0.3, 0.4)
    assert abs(phys_ref.hull_distance(unit, unit + [1.1, 1.2, 1.2]) - 0.3) < 1e-7           # corner to corner: (0.1, 0.2, 0.2)
    assert phys_ref.hull_distance(unit, unit + [0.5, 0.5, 0.5]) < 1e-9                      # overlapping
    tet = np.array([[2.0, 0.5, 0.5], [3.0, 0.0, 0.0], [3.0, 1.0, 0.0], [3.0, 0.5, 1.0]])
    assert abs(phys_ref.hull_distance(unit, tet) - 1.0) < 1e-7                              # vertex to face
    # margin 1 mm per shape: contact below a 2 mm gap
    assert phys_ref.hulls_intersect(unit, unit + [1.0015, 0, 0], margin=0.001)
    assert not phys_ref.hulls_intersect(unit, unit + [1.0025, 0, 0], margin=0.001)
    assert not phys_ref.hulls_intersect(unit, unit + [1.0015, 0, 0])                       # plain intersection: apart
    c = (box([-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]) @ rot_z(np.pi / 4).T)                     # edge at x = 0.70711
    assert phys_ref.hulls_intersect(c, unit + [0.7085, -0.5, -0.5], margin=0.001)
    assert not phys_ref.hulls_intersect(c, unit + [0.7095, -0.5, -0.5], margin=0.001)


def _write_obj(path, parts, header="", group="o"):
    """parts: list of (vertices [V,3], faces as 0-based local index triples)"""
    lines, base = [header] if header else [], 0
    for k, (v, faces) in enumerate(parts):
        lines.append(f"{group} part_{k}")
        lines += [f"v {x:.9g} {y:.9g} {z:.9g}" for x, y, z in v]
        lines += ["f " + " ".join(str(base + i + 1) for i in f) for f in faces]
        base += len(v)
    open(path, "w").write("\n".join(lines) + "\n")


BOX_FACES = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]


def test_hulls_from_obj_splits_shapes_like_the_mesh_loader(tmp_path):
    """PyBullet's GEOM_MESH loader makes one convex hull per shape of the .obj (reference vision_3d/physics_utils.py:238
    loads obj.phys_model that way; VHACD writes one `o` group per convex part)."""
    from dream2real_amd import physics_utils
    a, b = box([0, 0, 0], [1, 1, 1]), box([2, 0, 0], [3, 1, 0.5])
    p = str(tmp_path / "two.obj")
    _write_obj(p, [(a, BOX_FACES), (b, BOX_FACES)], header="# VHACD-style output")
    hulls = physics_utils.hulls_from_obj(p)
    assert len(hulls) == 2
    np.testing.assert_allclose(hulls[0], a) and np.testing.assert_allclose(hulls[1], b)
    _write_obj(p, [(a, BOX_FACES), (b, BOX_FACES)], group="g")
    assert len(physics_utils.hulls_from_obj(p)) == 2
    # no groups: one shape; vertices no face references are not part of it; v/vt/vn and negative indices
    lines = [f"v {x} {y} {z}" for x, y, z in a] + ["v 9 9 9", "vt 0 0", "vn 0 0 1"]
    lines += ["f " + " ".join(f"{i + 1}/1/1" for i in f) for f in BOX_FACES[:6]] + ["f -3/1/1 -2/1/1 -4/1/1"]
    open(p, "w").write("\n".join(lines) + "\n")
    (h,) = physics_utils.hulls_from_obj(p)
    assert len(h) == 8 and not (h == 9).any()
    # a group without faces is skipped; a file with vertices only is one hull of all of them
    open(p, "w").write("o empty\nv 5 5 5\no real\n" + "\n".join(f"v {x} {y} {z}" for x, y, z in a) + "\nf 2 3 4\n")
    (h,) = physics_utils.hulls_from_obj(p)
    assert len(h) == 3
    open(p, "w").write("\n".join(f"v {x} {y} {z}" for x, y, z in b) + "\n")
    (h,) = physics_utils.hulls_from_obj(p)
    np.testing.assert_allclose(h, b)
    open(p, "w").write("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        physics_utils.hulls_from_obj(p)
    # object_hulls: vertex arrays take precedence over the mesh path
    import types
    _write_obj(p, [(a, BOX_FACES)])
    assert len(physics_utils.object_hulls(types.SimpleNamespace(phys_model=p))) == 1
    assert len(physics_utils.object_hulls(types.SimpleNamespace(phys_model=p, phys_hulls=[a, b, a]))) == 3
    with pytest.raises(ValueError):
        physics_utils.object_hulls(types.SimpleNamespace(phys_model=None))


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


BAND = 5e-6      # metres of collision margin = 1e-5 m of hull distance


def assert_equal_away_from_band(got, want_fn, margin, what=""):
    """The GPU decides contact by float32 distance GJK, the oracle by an LP / QP in double: a pair whose hull distance lies
    within BAND-rounding of the contact distance 2 * margin may fall either way; everywhere else the masks must be EQUAL.
    Checked as: every pose's GPU answer equals the oracle's answer for the margin itself or for a margin BAND smaller or
    larger (a contact distance within +-1e-5 m)."""
    w0 = want_fn(margin)
    ok = got == w0
    n_off = int((~ok).sum())
    if n_off:
        ok |= got == want_fn(max(0.0, margin - BAND))
        ok |= got == want_fn(margin + BAND)
    print(f"[parity] physics {what} margin {margin}: {n_off} of {len(got)} poses differ from the oracle at the margin itself, "
          f"{int((~ok).sum())} outside the +-{2 * BAND:.0e} m band")
    assert ok.all(), (what, margin, np.nonzero(~ok)[0][:10])
    return w0


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
        want = assert_equal_away_from_band(got, lambda m: phys_ref.unsupcol_check(poses, init, mov, [shelf, blob], res, v0, 0.2, disallow_regrasp=regrasp, margin=m),
                                           0.0, f"six-DoF grid, regrasp={regrasp}")
        assert (got & ~v0).sum() == 0
    assert want.sum() > 0
    # (3) the closure optimise_pose_grid takes as phys_check
    import types, torch
    task = types.SimpleNamespace(movable_obj=types.SimpleNamespace(pose=torch.tensor(init), phys_hull=mov),
                                 task_bground_obj=types.SimpleNamespace(phys_hulls=[shelf, blob]),
                                 scene_model=types.SimpleNamespace(scene_centre=torch.tensor([0.45, 0.85, 0.20])))
    check, static_handles, movable_handles = physics_utils.create_unsupcol_check(ctx, task, res, embodied=False, margin=0.0)
    out = check(torch.from_numpy(poses), task, torch.from_numpy(v0))
    assert out.dtype == torch.bool and (out.numpy() == sh.check(poses, v0, res, init, 0.2)).all()
    assert len(static_handles) == 1 and len(static_handles[0]) == 2 and len(movable_handles) == 1      # the reference's triple (:377)
    check.shapes.close(); sh.close(); ctx.close()


@pytest.mark.gpu
def test_gpu_prefilter_with_margins_compound_parts_and_mesh_files(tmp_path):
    """The drop-in form: shapes from .obj files (one hull per `o` group, the movable object itself a compound of two
    convex parts), lazy_phys_mods on and off (reference vision_3d/physics_utils.py:235), and contact with PyBullet's
    collision margin (believed 1 mm per convex part) against the oracle's distance test, margin 0 and > 0."""
    import types
    import torch
    from dream2real_amd import engine, physics_utils
    ctx = engine.Context(0)
    table = box([-1, -1, -0.1], [1, 1, 0.0])
    block = box([0.30, -0.10, 0.0], [0.50, 0.10, 0.20])
    post = box([0.70, -0.03, 0.0], [0.74, 0.03, 0.05])                      # a pebble low enough to fit under the plate
    # a movable object of two convex parts, a plate on one leg (its convex hull would fill the space under the plate)
    part_a = box([-0.05, -0.05, 0.08], [0.05, 0.05, 0.12]) + [0, 0, 0.005]
    part_b = box([-0.05, -0.05, 0.0], [-0.01, 0.05, 0.08]) + [0, 0, 0.005]
    init = np.eye(4, dtype=np.float32)
    # (the grid is offset by a millimetre so that no face lies exactly in the plane of another: at distance exactly 0 an
    # LP says "they share a point" while GJK's progress test says "no closer than touching" — either is right)
    xs, ys, zs = np.linspace(-0.6, 1.15, 36) + 0.0013, [0.0007, 0.0707], [0.0, 0.0115, 0.0135, 0.1, 0.2055, 0.2175, 0.2195, -0.5]
    poses = _grid(xs, ys, zs)
    res = [len(xs), len(ys), len(zs), 1, 1, 1]
    v0 = np.ones(len(poses), bool)
    # (1) margins on explicit hulls: 5 mm initial gap + lowering by 2 cm; z offsets 0.0115 / 0.0135 put the lowered part
    # 1.5 mm / 3.5 mm... above the table: inside / outside the 2 mm contact distance
    sh = physics_utils.PhysicsShapes(ctx, [part_a, part_b], [table, block, post])
    results = {}
    for m in (0.0, 0.001, 0.004):
        got = sh.check(poses, v0, res, init, -0.3, margin=m)
        want = assert_equal_away_from_band(got, lambda mm: phys_ref.unsupcol_check(poses, init, [part_a, part_b], [table, block, post], res, v0, -0.3, margin=mm),
                                           m, "compound movable object")
        results[m] = want
    assert (results[0.0] != results[0.001]).any() and (results[0.001] != results[0.004]).any()     # the margin matters on this grid
    # the compound is not its hull: with the pebble under the plate beside the leg a pose is valid only as two parts
    hull_only = phys_ref.unsupcol_check(poses, init, np.concatenate([part_a, part_b]), [table, block, post], res, v0, -0.3)
    assert (hull_only != results[0.0]).any()
    sh.close()
    # (2) the same shapes through mesh files and the reference's call: lazy_phys_mods=True -> merged background + movable
    bg_path, mov_path = str(tmp_path / "bground.obj"), str(tmp_path / "movable.obj")
    _write_obj(bg_path, [(table, BOX_FACES), (block, BOX_FACES), (post, BOX_FACES)])
    _write_obj(mov_path, [(part_a, BOX_FACES), (part_b, BOX_FACES)])
    movable = types.SimpleNamespace(pose=torch.tensor(init), phys_model=mov_path)
    bground = types.SimpleNamespace(phys_model=bg_path)
    scene_model = types.SimpleNamespace(scene_centre=torch.tensor([0.0, 0.0, -0.3]), objs=None)
    task = types.SimpleNamespace(movable_obj=movable, task_bground_obj=bground, scene_model=scene_model)
    check, static_handles, movable_handles = physics_utils.create_unsupcol_check(ctx, task, res, embodied=False, lazy_phys_mods=True)
    out = check(torch.from_numpy(poses), task, torch.from_numpy(v0)).numpy()
    assert_equal_away_from_band(out, lambda mm: results[mm] if mm in results else phys_ref.unsupcol_check(
        poses, init, [part_a, part_b], [table, block, post], res, v0, -0.3, margin=mm), physics_utils.PYBULLET_MESH_MARGIN, "mesh files, lazy_phys_mods")
    assert [len(p) for p in static_handles] == [3] and [len(p) for p in movable_handles] == [2]
    check.shapes.close()
    # lazy_phys_mods=False: every scene object is its own body, all but the movable one static (:235-245)
    paths = []
    for k, h in enumerate((table, block, post)):
        paths.append(str(tmp_path / f"obj{k}.obj"))
        _write_obj(paths[-1], [(h, BOX_FACES)])
    scene_model.objs = [types.SimpleNamespace(phys_model=paths[0]), movable, types.SimpleNamespace(phys_model=paths[1]),
                        types.SimpleNamespace(phys_model=paths[2])]
    check2, static2, movable2 = physics_utils.create_unsupcol_check(ctx, task, res, embodied=False, lazy_phys_mods=False)
    out2 = check2(torch.from_numpy(poses), task, torch.from_numpy(v0)).numpy()
    np.testing.assert_array_equal(out2, out)
    assert len(static2) == 3 and len(movable2) == 1
    check2.shapes.close()
    ctx.close()


@pytest.mark.gpu
def test_gpu_prefilter_on_flat_hulls_and_duplicate_vertices():
    """Degenerate simplices: a movable object that is a flat sheet (every vertex in one plane, each vertex
    listed twice, collinear points on its edges) against a table and a thin blade — GJK then meets collinear / repeated
    support points, where the interior-of-triangle formula divides by zero.  The mask must still equal the oracle's
    (LP / QP on the same point sets) outside the rounding band, and contain no pose that a NaN would have let through."""
    from dream2real_amd import engine, physics_utils
    ctx = engine.Context(0)
    table = box([-1, -1, -0.1], [1, 1, 0.0])
    blade = np.array([[0.4, -0.2, 0.0], [0.4, 0.2, 0.0], [0.4, 0.2, 0.3], [0.4, -0.2, 0.3],          # a zero-thickness wall, x = 0.4
                      [0.4, 0.0, 0.0], [0.4, 0.0, 0.3], [0.4, -0.2, 0.15]], np.float64)            # + collinear points on its edges
    sq = np.array([[-0.05, -0.05, 0.0], [0.05, -0.05, 0.0], [0.05, 0.05, 0.0], [-0.05, 0.05, 0.0],
                   [0.0, -0.05, 0.0], [0.05, 0.0, 0.0], [0.0, 0.0, 0.0]], np.float64) + [0, 0, 0.005]
    sheet = np.concatenate([sq, sq])                                                                  # every vertex twice
    init = np.eye(4, dtype=np.float32)
    xs, ys, zs = np.linspace(0.2, 0.6, 41) + 0.00037, [0.0003, 0.1003, 0.2603], [0.0, 0.0125, 0.0175, 0.1, -0.5]
    poses = _grid(xs, ys, zs)
    res = [len(xs), len(ys), len(zs), 1, 1, 1]
    v0 = np.ones(len(poses), bool)
    sh = physics_utils.PhysicsShapes(ctx, sheet, [table, blade])
    for m in (0.0, 0.001):
        for stab in (False, True):
            got = sh.check(poses, v0, res, init, -0.3, margin=m, stability_check=stab)
            want = assert_equal_away_from_band(got, lambda mm: phys_ref.unsupcol_check(poses, init, sheet, [table, blade], res, v0, -0.3, margin=mm,
                                                                                   stability_check=stab), m, f"flat hulls, stability={stab}")
            assert 0 < want.sum() < len(want)
    sh.close()
    ctx.close()
