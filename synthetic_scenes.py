"""Seeded synthetic scenes / tasks for bench.py, smoke() and the parity tests.

The reference consumes trained instant-ngp snapshots (`fg_base.ingp`, `bg_base.ingp`, reference
reconstruction/ngp_visual_model.py:20-29).  None is available offline, so tests and benchmarks run on
seeded synthetic models with the same structure.  Fixtures only: they build *inputs* that are handed
unchanged to both the HIP library and the oracle; the product package does not import this module
(tests/scenes.py re-exports it for the test suite).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

from dream2real_amd.scene import DEMO_LENS, GRID, LENS_OPENCV, GridLevels, NerfModel, View, grid_levels, pack_bits, world_to_ngp


def _xavier(rng: np.random.Generator, n_out: int, n_in: int) -> np.ndarray:
    lim = math.sqrt(6.0 / (n_in + n_out))
    return rng.uniform(-lim, lim, size=(n_out, n_in)).astype(np.float32)


def make_synthetic_nerf(occ_zyx: np.ndarray, *, seed_grid: int, seed_mlp: int,
                        levels: Optional[GridLevels] = None, log_sigma: float = 5.0,
                        aabb_scale: int = 1) -> NerfModel:
    """Seeded random NeRF with a controllable opacity.

    Tables are U(-0.5, 0.5) (PCG64); feature 0 of level 0 is pinned to 0.5 so that a
    dedicated hidden unit carries a constant through the density MLP and
    sigma = exp(out0) ~ exp(log_sigma): occupied space is opaque within ~8-20 samples,
    as a trained object would be, while colour and fine structure stay random."""
    levels = levels or grid_levels()
    F = levels.n_features
    rg = np.random.Generator(np.random.PCG64(seed_grid))
    grid = rg.uniform(-0.5, 0.5, size=(levels.n_entries, F)).astype(np.float32)
    grid[levels.offset[0]:levels.offset[0] + levels.size[0], 0] = 0.5
    rm = np.random.Generator(np.random.PCG64(seed_mlp))
    n_in = levels.n_levels * F
    dw1 = _xavier(rm, 64, n_in)
    dw2 = _xavier(rm, 16, 64)
    cw1 = _xavier(rm, 64, 32)
    cw2 = _xavier(rm, 64, 64)
    cw3 = _xavier(rm, 16, 64)
    dw1[0, :] = 0.0
    dw1[0, 0] = 2.0            # hidden0 = relu(2 * 0.5) = 1
    dw2[0, 0] = log_sigma      # out0 = log_sigma + noise
    f16 = lambda a: a.astype(np.float16)
    return NerfModel(levels, f16(grid), f16(dw1), f16(dw2), f16(cw1), f16(cw2), f16(cw3),
                     pack_bits(occ_zyx), aabb_scale)


def make_trained_like_nerf(occ_zyx: np.ndarray, centre_ngp, radii_ngp, *, seed_grid: int, seed_mlp: int,
                            levels: Optional[GridLevels] = None, peak: float = 12.0, fall_per_metre: float = 2400.0,
                            table_limit: float = 8.0, mlp_gain: float = 1.6, aabb_scale: int = 1) -> NerfModel:
    """A field with the statistics of a TRAINED instant-ngp object rather than of a random one:
    * hash-table values heavy-tailed (Student-t, 3 degrees of freedom) and clipped to +-`table_limit` (a trained table
      spans several units where initialisation is 1e-4);
    * MLP weights `mlp_gain` x Xavier;
    * density pre-activation out0 from +`peak` ON the surface of the ellipsoid (centre / radii in ngp units) down to
      -`peak` at 1 cm from it and below further away: sigma = exp(out0) spans exp(+-12) = 1.6e5 ... 6e-6 and the
      opaque region is a SHELL ~ 3 march steps thick (out0 = peak - fall_per_metre * |distance|), carried by a signed
      distance stored in feature 0 of the finest DENSE level and read by two ReLU units of opposite sign.
    Same structure and file layout as make_synthetic_nerf; inputs for oracle and library alike."""
    levels = levels or grid_levels()
    F = levels.n_features
    rg = np.random.Generator(np.random.PCG64(seed_grid))
    grid = np.clip(rg.standard_t(3, size=(levels.n_entries, F)) * 0.7, -table_limit, table_limit).astype(np.float32)
    grid[levels.offset[0]:levels.offset[0] + levels.size[0], 0] = 0.5
    dense = np.nonzero(~levels.hashed)[0]
    ls = int(dense[-1])                                         # the finest level that is indexed densely
    res, scale = int(levels.res[ls]), float(levels.scale[ls])
    g = (np.arange(res, dtype=np.float64) - 0.5) / scale        # vertex g of a level sits at x = (g - 0.5) / scale (pos = x * scale + 0.5)
    gz, gy, gx = np.meshgrid(g, g, g, indexing="ij")
    c, r = np.asarray(centre_ngp, np.float64), np.asarray(radii_ngp, np.float64)
    q = np.sqrt(((gx - c[0]) / r[0]) ** 2 + ((gy - c[1]) / r[1]) ** 2 + ((gz - c[2]) / r[2]) ** 2)
    sdf = (q - 1.0) * r.min()                                   # ~ signed distance to the ellipsoid (exact for a sphere), metres = ngp units
    G = table_limit / 0.02                                      # the carrier saturates 2 cm from the surface
    carrier = np.clip(sdf * G, -table_limit, table_limit).astype(np.float32).reshape(-1)     # index x + res * (y + res * z)
    grid[levels.offset[ls]:levels.offset[ls] + res ** 3, 0] = carrier
    rm = np.random.Generator(np.random.PCG64(seed_mlp))
    n_in = levels.n_levels * F
    dw1 = _xavier(rm, 64, n_in) * mlp_gain
    dw2 = _xavier(rm, 16, 64) * mlp_gain
    cw1 = _xavier(rm, 64, 32) * mlp_gain
    cw2 = _xavier(rm, 64, 64) * mlp_gain
    cw3 = _xavier(rm, 16, 64) * mlp_gain
    k = 2.0
    slope = fall_per_metre / (k * G)
    dw1[0:3, :] = 0.0
    dw1[0, 0] = 2.0                     # hidden0 = relu(2 * 0.5) = 1  (constant carrier, as in make_synthetic_nerf)
    dw1[1, ls * F] = k                  # hidden1 = relu(+k * carrier): outside the surface
    dw1[2, ls * F] = -k                 # hidden2 = relu(-k * carrier): inside
    dw2[0, 0] = peak
    dw2[0, 1] = -slope
    dw2[0, 2] = -slope
    f16 = lambda a: a.astype(np.float16)
    return NerfModel(levels, f16(grid), f16(dw1), f16(dw2), f16(cw1), f16(cw2), f16(cw3),
                     pack_bits(occ_zyx), aabb_scale)


def _cell_centres(cascade: int = 0):
    """ngp-space centres of the 128^3 cells of an occupancy cascade (side 2^cascade about 0.5)."""
    side = float(1 << cascade)
    c = (np.arange(GRID, dtype=np.float64) + 0.5) / GRID * side + 0.5 - side / 2
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    return x, y, z


def ellipsoid_occupancy(centre_ngp, radii_ngp, cascade: int = 0) -> np.ndarray:
    x, y, z = _cell_centres(cascade)
    cx, cy, cz = centre_ngp
    rx, ry, rz = radii_ngp
    return ((x - cx) / rx) ** 2 + ((y - cy) / ry) ** 2 + ((z - cz) / rz) ** 2 <= 1.0


def look_at_opencv(eye, target, up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """Camera-to-world 4x4 in the OpenCV convention (x right, y down, z forward) — the
    convention of the reference's opt_cam_poses before utils/accio2ngp.py:133 flips it."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = right, down, fwd, eye
    return T


@dataclasses.dataclass
class SyntheticScene:
    """Everything optimise_pose_grid needs from a TaskModel, for a seeded synthetic task."""
    name: str
    scene_type: int
    scene_centre: np.ndarray      # world
    fg: NerfModel
    bg: NerfModel
    obj_pose: np.ndarray          # T_WO_1, 4x4 world pose of the movable object now
    cam_poses: np.ndarray         # [V,4,4] opt_cam_poses (OpenCV convention)
    fg_background: tuple          # Testbed.background_color of the fg model
    lens: Optional[tuple] = None  # OpenCV (k1, k2, p1, p2) of the training views (scene.DEMO_LENS = the reference's configs), None = pinhole

    def view(self, width: int, height: int) -> View:
        """The view a Testbed of this scene renders with after set_camera_to_training_view(0)."""
        if self.lens is not None and any(self.lens):
            return View.from_training_view(width, height, lens_mode=LENS_OPENCV, lens_params=tuple(float(x) for x in self.lens))
        return View.from_training_view(width, height)

    @property
    def training_views(self):
        """`training_views` of engine.Testbed for this scene: the RealSense intrinsics of configs/shopping_demo.json:49-60
        and the scene's lens."""
        return [dict(fx=924.66912, fy=926.49735, cx=654.51953, cy=355.18523, w=1280, h=720, lens=self.lens)]

    def testbeds(self, ctx):
        """(fg, bg) engine.Testbed of the scene, their camera set to training view 0 as the reference's renderer does before
        every render (reconstruction/combined_rendering.py:98,116) — which also makes the view's lens the render lens."""
        from dream2real_amd import engine
        fg = engine.Testbed(ctx, self.fg, training_views=self.training_views)
        bg = engine.Testbed(ctx, self.bg, training_views=self.training_views)
        fg.background_color = list(self.fg_background)
        for tb in (fg, bg):
            tb.set_camera_to_training_view(0)
        return fg, bg


def make_scene(kind: str = "shopping", lens=None) -> SyntheticScene:
    sc = _make_scene(kind)
    sc.lens = tuple(lens) if lens is not None else None
    return sc


def _make_scene(kind: str = "shopping") -> SyntheticScene:
    """Seeded scenes of SURVEY.md §8(d).  kind: 'shopping' (apple-sized ellipsoid, scene
    type 3), 'pool_triangle' (2.8 cm sphere, scene type 0) or 'shelf' (aabb_scale 2 like
    configs/shelf_demo.json:62: two occupancy cascades, cone stepping; the object sits outside
    the unit cube and the camera 1.3 m away).  'shopping_trained': the shopping scene with a foreground
    field of trained-like statistics (make_trained_like_nerf: thin opaque shell, sigma over exp(+-12),
    table values to +-8)."""
    trained = kind.endswith("_trained")
    if trained:
        kind = kind[:-len("_trained")]
    if kind == "shelf":
        return _make_shelf_scene()
    if kind == "room":
        return _make_shelf_scene(aabb_scale=4)
    levels = grid_levels()
    scene_centre = np.array([0.5, 0.0, 0.035])          # configs/shopping_demo.json:29
    if kind in ("shopping", "shopping_big", "shopping_huge"):
        # _big / _huge: the same scene with an object 2.2x / 5x the apple's size (bench.py's marcher-regime sweep: the
        # level bricks of such an object no longer fit five LDS slots / only the coarsest do)
        k = {"shopping": 1.0, "shopping_big": 2.2, "shopping_huge": 5.0}[kind]
        scene_type, radii_w = 3, (0.04 * k, 0.04 * k, 0.05 * k)
        obj_t = scene_centre + np.array([-0.02, -0.05, 0.05 * k])
    elif kind == "pool_triangle":
        scene_type, radii_w = 0, (0.028, 0.028, 0.028)
        obj_t = scene_centre + np.array([-0.03, -0.02, 0.028])
    else:
        raise ValueError(kind)
    obj_pose = np.eye(4)
    obj_pose[:3, 3] = obj_t
    # world radii (x,y,z) -> ngp axes (y,z,x)
    radii_ngp = (radii_w[1], radii_w[2], radii_w[0])
    fg_occ = ellipsoid_occupancy(world_to_ngp(obj_t), radii_ngp)
    if trained:
        fg = make_trained_like_nerf(fg_occ, world_to_ngp(obj_t), radii_ngp, seed_grid=1, seed_mlp=3, levels=levels)
    else:
        fg = make_synthetic_nerf(fg_occ, seed_grid=1, seed_mlp=3, levels=levels)
    # background: table slab (6 cm under world z=0, i.e. ngp y in [0.44,0.5]) + three blobs
    x, y, z = _cell_centres()
    bg_occ = (y >= 0.44) & (y < 0.5)
    for dx, dy, r in ((0.12, -0.10, 0.05), (-0.15, 0.08, 0.04), (0.05, 0.15, 0.045)):
        c = world_to_ngp(scene_centre + np.array([dx, dy, r - 0.035]))
        bg_occ |= ellipsoid_occupancy(c, (r, r, r))
    bg = make_synthetic_nerf(bg_occ, seed_grid=2, seed_mlp=4, levels=levels)
    eye = scene_centre + 0.6 * np.array([-0.35, -0.45, 0.82]) / np.linalg.norm([-0.35, -0.45, 0.82])
    cams = np.stack([look_at_opencv(eye, scene_centre),
                     look_at_opencv(eye + np.array([0.1, 0.0, 0.02]), scene_centre)])
    return SyntheticScene(kind + ("_trained" if trained else ""), scene_type, scene_centre, fg, bg, obj_pose, cams,
                          fg_background=(0.0, 0.0, 0.0, 1.0))


def _make_shelf_scene(aabb_scale: int = 2) -> SyntheticScene:
    """aabb_scale 2: the 'shelf' scene.  aabb_scale 4 ('room'): the same geometry in a box of side 4 with three
    occupancy cascades, a far wall that only cascade 2 holds, and the camera 2.3 m away (steps grow past the
    cascade thresholds at t = 1 and t = 2)."""
    levels = grid_levels(aabb_scale=aabb_scale)
    n_casc = int(aabb_scale).bit_length()
    scene_centre = np.array([0.45, 0.85, 0.20])                 # world; ngp (1.15, 0.70, 0.45): outside the unit cube
    obj_t = scene_centre + np.array([0.02, -0.03, 0.06])
    obj_pose = np.eye(4)
    obj_pose[:3, 3] = obj_t
    radii_ngp = (0.06, 0.08, 0.06)

    def cascades(fn):
        # as instant-ngp builds its bitfield: cascade 1 also holds the 2x2x2 max-pool of cascade 0 in its
        # central half (dream2real_amd.ingp.occupancy_from_density), so a snapshot round trip is the identity
        h, q = GRID // 2, GRID // 4
        out = [fn(0)]
        for c in range(1, n_casc):
            cc = fn(c).copy()
            cc[q:q + h, q:q + h, q:q + h] |= out[-1].reshape(h, 2, h, 2, h, 2).any(axis=(1, 3, 5))
            out.append(cc)
        return np.stack(out)
    fg = make_synthetic_nerf(cascades(lambda c: ellipsoid_occupancy(world_to_ngp(obj_t), radii_ngp, c)),
                             seed_grid=1, seed_mlp=3, levels=levels, aabb_scale=aabb_scale)

    def bg_occ(c):
        x, y, z = _cell_centres(c)
        occ = (y >= 0.50) & (y < 0.58) & (x > -0.2) & (x < 1.3)                      # a shelf board
        occ |= (z >= 1.20) & (z < 1.28) & (y > 0.3) & (y < 1.2)                      # the back panel
        for d, r in (((0.20, -0.15, 0.05), 0.09), ((-0.25, 0.10, 0.02), 0.07)):
            occ |= ellipsoid_occupancy(world_to_ngp(scene_centre + np.array(d)), (r, r, r), c)
        if aabb_scale > 2:
            occ |= (z >= 1.95) & (z < 2.10) & (y > -0.5) & (y < 1.8) & (x > -1.0) & (x < 2.0)      # a far wall: outside cascades 0 and 1
        return occ
    bg = make_synthetic_nerf(cascades(bg_occ), seed_grid=2, seed_mlp=4, levels=levels, aabb_scale=aabb_scale)
    dist = 1.3 if aabb_scale == 2 else 2.3
    eye = scene_centre + dist * np.array([-0.55, -0.35, 0.75]) / np.linalg.norm([-0.55, -0.35, 0.75])
    cams = np.stack([look_at_opencv(eye, scene_centre),
                     look_at_opencv(eye + np.array([0.15, 0.0, 0.05]), scene_centre)])
    return SyntheticScene("shelf" if aabb_scale == 2 else "room", 1, scene_centre, fg, bg, obj_pose, cams,
                          fg_background=(0.0, 0.0, 0.0, 1.0))


def scene_text_embeds(image_embed, n_caps: int = 2, seed: int = 5, noise: float = 0.8) -> np.ndarray:
    """Cached "caption" embeddings for a synthetic task: seeded unit vectors positively correlated
    with an image embedding of the scene (as a real goal/normalising caption pair would be), so the
    logits are positive and the goal/norm ratio is well conditioned.  Input data only."""
    e = np.asarray(image_embed, np.float64).reshape(-1)
    r = np.random.Generator(np.random.PCG64(seed))
    t = e[None] + noise * r.standard_normal((n_caps, e.size)) / np.sqrt(e.size) * np.linalg.norm(e)
    return (t / np.linalg.norm(t, axis=-1, keepdims=True)).astype(np.float32)


def make_task(scene: SyntheticScene, fg_tb=None, bg_tb=None):
    """Duck-typed TaskModel (reference scene_model.py:45-130) for a synthetic scene: the fields
    optimise_pose_grid / renderer read."""
    import types

    import torch
    sm = types.SimpleNamespace(scene_centre=torch.tensor(scene.scene_centre, dtype=torch.float32),
                               opt_cam_poses=[torch.tensor(p, dtype=torch.float32) for p in scene.cam_poses],
                               device="cpu")
    return types.SimpleNamespace(
        scene_model=sm,
        movable_obj=types.SimpleNamespace(vis_model=fg_tb, pose=torch.tensor(scene.obj_pose, dtype=torch.float32)),
        task_bground_obj=types.SimpleNamespace(vis_model=bg_tb),
        goal_caption="an apple inside a blue and white bowl",
        norm_captions=["an apple and a blue and white bowl"],
        movable_masks=None)


# ----------------------------------------------------------------------------------------------------------------
# Physics shapes of the synthetic scenes, as the reference's mesh pipeline would have written them (get_phys_models:
# one .obj per object, world coordinates, one `o` group per convex part) — inputs of the physics pre-filter
# (reference vision_3d/physics_utils.py:232-375).

def box(lo, hi) -> np.ndarray:
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])


def icosphere(centre, r, n=40, seed=0) -> np.ndarray:
    g = np.random.default_rng(seed).standard_normal((n, 3))
    return np.asarray(centre) + r * g / np.linalg.norm(g, axis=1, keepdims=True)


def scene_phys_hulls(scene: SyntheticScene):
    """(movable parts, static parts) of the 'shopping' / 'pool_triangle' scenes: the object as a 40-vertex sphere hull,
    the table slab and the three background blobs."""
    c = np.asarray(scene.scene_centre, np.float64)
    table = box([c[0] - 0.6, c[1] - 0.6, -0.06], [c[0] + 0.6, c[1] + 0.6, 0.0])
    blobs = [icosphere(c + np.array([dx, dy, r - 0.035]), r, 40, k)
             for k, (dx, dy, r) in enumerate(((0.12, -0.10, 0.05), (-0.15, 0.08, 0.04), (0.05, 0.15, 0.045)))]
    radius = 0.03 if scene.name == "shopping" else 0.025
    apple = icosphere(np.asarray(scene.obj_pose)[:3, 3], radius, 40, 9)
    return [apple], [table] + blobs


def write_phys_meshes(scene: SyntheticScene, out_dir: str):
    """Writes bground.obj (every static part as its own `o` group: lazy_phys_mods' merged background object) and
    movable.obj; -> (movable path, background path)."""
    import os
    movable, static = scene_phys_hulls(scene)
    bg_path, mov_path = os.path.join(out_dir, "bground.obj"), os.path.join(out_dir, "movable.obj")
    with open(bg_path, "w") as f:
        base = 0
        for k, h in enumerate(static):
            f.write(f"o part_{k}\n" + "".join(f"v {x:.9g} {y:.9g} {z:.9g}\n" for x, y, z in h))
            f.write("".join(f"f {base + i + 1} {base + i + 2} {base + i + 3}\n" for i in range(len(h) - 2)))   # any faces that reference every vertex
            base += len(h)
    with open(mov_path, "w") as f:
        f.write("".join(f"v {x:.9g} {y:.9g} {z:.9g}\n" for x, y, z in movable[0]))
    return mov_path, bg_path
