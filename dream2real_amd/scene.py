"""Scene description for the render-and-score path: hash-grid metadata, NeRF parameter
containers and the camera/view state.  (The seeded synthetic scenes the tests and bench.py run on
live in synthetic_scenes.py: they are fixtures, not product.)

The reference consumes trained instant-ngp snapshots (`fg_base.ingp`, `bg_base.ingp`,
reference reconstruction/ngp_visual_model.py:20-29).  None is available offline, so the
benchmarks and parity tests run on seeded synthetic models with the same structure
(SURVEY.md §8(d)): a 16-level, 2-feature, 2^19-entry hash grid with fp16 tables, a
32->64->16 density MLP, a 32->64->64->16 colour MLP, and a 128^3 occupancy bitfield.

Nothing here is arithmetic of the hot path: these are the containers of its *inputs* (tables,
weights, bitfields, cameras), handed unchanged to the HIP library.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

GRID = 128          # occupancy grid side (instant-ngp NERF_GRIDSIZE)
LENS_PERSPECTIVE, LENS_OPENCV = 0, 1      # d2r.h D2R_LENS_*
# the OpenCV coefficients (k1, k2, p1, p2) every demo config of the reference writes into the transforms its NeRFs are
# trained from (configs/shopping_demo.json:51-56, shelf_demo.json:51-56; reconstruction/train_ngp.py:171-180)
DEMO_LENS = (0.096692, -0.166479, -0.000194, 0.002049)
DT = np.float32(math.sqrt(3.0) / 1024.0)   # constant march step at aabb_scale 1


@dataclasses.dataclass
class GridLevels:
    """Per-level constants of a tiny-cuda-nn style multiresolution hash grid."""
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    scale: np.ndarray      # [L] float32
    res: np.ndarray        # [L] uint32
    size: np.ndarray       # [L] uint32, entries per level
    offset: np.ndarray     # [L] uint32, first entry of each level
    n_entries: int

    @property
    def hashed(self) -> np.ndarray:
        """True where the level needs the spatial hash (res^3 exceeds the level size)."""
        return (self.res.astype(np.uint64) ** 3) > self.size.astype(np.uint64)


def grid_levels(n_levels: int = 16, n_features: int = 2, log2_hashmap_size: int = 19,
                base_resolution: int = 16, per_level_scale: Optional[float] = None,
                aabb_scale: int = 1) -> GridLevels:
    """Level table exactly as tiny-cuda-nn's GridEncoding constructor derives it
    (float32 exp2f/ceilf; sizes rounded up to 8 and capped at 2^log2_hashmap_size)."""
    if per_level_scale is None:
        # instant-ngp: desired resolution 2048*aabb_scale at the finest level
        per_level_scale = math.exp(math.log(2048.0 * aabb_scale / base_resolution) / (n_levels - 1))
    # log2 / exp2 in double, rounded to float32 (= a correctly rounded log2f / exp2f; dream2real_amd/csrc/ingp.hip
    # d2r_grid_levels does the same, so the C-ABI snapshot loader derives the same table bit for bit)
    log2_pls = np.float32(math.log2(float(np.float32(per_level_scale))))
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.uint32)
    size = np.zeros(n_levels, np.uint32)
    offset = np.zeros(n_levels, np.uint32)
    off = 0
    for l in range(n_levels):
        s = np.float32(2.0 ** float(np.float32(l) * log2_pls)) * np.float32(base_resolution) - np.float32(1.0)
        r = int(math.ceil(float(s))) + 1
        max_params = (2 ** 32 - 1) // 2
        p = max_params if float(r) ** 3 > float(max_params) else r ** 3
        p = (p + 7) // 8 * 8
        p = min(p, 1 << log2_hashmap_size)
        scale[l], res[l], size[l], offset[l] = s, r, p, off
        off += p
    return GridLevels(n_levels, n_features, log2_hashmap_size, base_resolution,
                      float(per_level_scale), scale, res, size, offset, off)


@dataclasses.dataclass
class NerfModel:
    """Host-side parameters of one NeRF (the state a pyngp.Testbed holds after
    load_snapshot, reference reconstruction/ngp_visual_model.py:24-28)."""
    levels: GridLevels
    grid: np.ndarray       # [n_entries, F] float16
    dw1: np.ndarray        # [64, 32] float16   density layer 1
    dw2: np.ndarray        # [16, 64] float16   density layer 2
    cw1: np.ndarray        # [64, 32] float16   colour layer 1 (in = [density out | SH])
    cw2: np.ndarray        # [64, 64] float16
    cw3: np.ndarray        # [16, 64] float16   rows 0..2 = rgb
    occ_bits: np.ndarray   # [n_cascades * 128^3/8] uint8, bit (x + 128*(y + 128*z)) per cascade, LSB first
    aabb_scale: int = 1    # a power of two: the box is the cube of that side centred at 0.5; cascade c (0 .. log2) covers side 2^c
    render_aabb: Optional[tuple] = None   # Testbed.render_aabb (lo xyz, hi xyz, ngp coordinates); None = the whole box

    @property
    def n_cascades(self) -> int:
        return int(self.aabb_scale).bit_length()

    def occupancy_bool(self) -> np.ndarray:
        """[z, y, x] boolean view of the bitfield ([cascade, z, y, x] when aabb_scale > 1)."""
        b = np.unpackbits(self.occ_bits, bitorder="little").astype(bool)
        return b.reshape(GRID, GRID, GRID) if self.aabb_scale == 1 else b.reshape(self.n_cascades, GRID, GRID, GRID)


@dataclasses.dataclass
class View:
    """Camera/intrinsics state the path sets on a Testbed before render()
    (reference reconstruction/combined_rendering.py:98-105,116,123-130)."""
    width: int
    height: int
    focal: tuple            # pixels at (width, height): rel_focal * height
    center: tuple           # principal point, relative (cx/w, cy/h)
    scale: float = 1.0      # dataset scale  (configs/shopping_demo.json:63)
    offset: tuple = (0.0, 0.3, 0.5)   # dataset offset (configs/shopping_demo.json:64)
    background: tuple = (0.0, 0.0, 0.0, 1.0)
    min_transmittance: float = 0.01
    near_distance: float = 0.0
    # the render lens set_camera_to_training_view leaves behind (nerf.render_with_lens_distortion + nerf.render_lens):
    # 0 = perspective, 1 = OpenCV (k1, k2, p1, p2) — every ray's camera-space direction is undistorted iteratively
    lens_mode: int = 0
    lens_params: tuple = (0.0, 0.0, 0.0, 0.0)

    @staticmethod
    def from_training_view(width: int, height: int, fx: float = 924.66912, fy: float = 926.49735,
                           cx: float = 654.51953, cy: float = 355.18523, w0: int = 1280,
                           h0: int = 720, **kw) -> "View":
        """set_camera_to_training_view + render(w,h): relative focal length is taken over
        the image *height* (fov axis 1) and re-applied at the render height.
        Defaults are the RealSense intrinsics of configs/shopping_demo.json:49-60."""
        rel = (fx / h0, fy / h0)
        return View(width, height, (rel[0] * height, rel[1] * height), (cx / w0, cy / h0), **kw)


def pack_bits(occ_zyx: np.ndarray) -> np.ndarray:
    return np.packbits(occ_zyx.reshape(-1).astype(np.uint8), bitorder="little")


def world_to_ngp(p, scale: float = 1.0, offset=(0.0, 0.3, 0.5)) -> np.ndarray:
    """World point -> instant-ngp unit-cube coordinates: p*scale+offset, then (x,y,z)<-(y,z,x)."""
    q = np.asarray(p, np.float64) * scale + np.asarray(offset, np.float64)
    return np.stack([q[..., 1], q[..., 2], q[..., 0]], axis=-1)
