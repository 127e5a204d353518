"""Reader for instant-ngp `.ingp` NeRF snapshots.

The reference loads `fg_base.ingp` / `bg_base.ingp` through `pyngp.Testbed.load_snapshot`
(reference reconstruction/ngp_visual_model.py:24-28).  No snapshot and no instant-ngp source is
available offline, so this module is written against the format as believed (SURVEY.md §3.4 and
Appendix A) and is exercised on snapshots produced by tests/ingp_writer.py in the same layout.
UNPINNED against real files: every constant is taken from the snapshot's embedded config, and
anything unexpected raises instead of guessing.

Believed layout: zlib/gzip-compressed msgpack of the network config JSON plus
  snapshot.params_binary        fp16: density MLP, colour MLP, then hash-grid tables
                                (tiny-cuda-nn FullyFusedMLP matrices row-major [out][in];
                                first layer [64][in_padded], hidden [64][64], last [16][64])
  snapshot.density_grid_binary  fp16 128^3 per cascade, Morton order
  snapshot.nerf.dataset         per-image metadata (focal length, principal point, resolution),
                                scale, offset, aabb_scale
"""
from __future__ import annotations

import gzip
import zlib

import numpy as np

from .scene import GRID, NerfModel, grid_levels

NERF_MIN_OPTICAL_THICKNESS = 0.01


def _morton_order() -> np.ndarray:
    """linear index x + 128*(y + 128*z) of every Morton code 0..128^3-1."""
    def compact(v):
        v = v & 0x09249249
        v = (v ^ (v >> 2)) & 0x030C30C3
        v = (v ^ (v >> 4)) & 0x0300F00F
        v = (v ^ (v >> 8)) & 0xFF0000FF
        v = (v ^ (v >> 16)) & 0x000003FF
        return v
    m = np.arange(GRID ** 3, dtype=np.uint32)
    x, y, z = compact(m), compact(m >> 1), compact(m >> 2)
    return (x + GRID * (y + GRID * z)).astype(np.int64)


def _decompress(raw: bytes) -> bytes:
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    try:
        return zlib.decompress(raw)
    except zlib.error:
        return raw          # uncompressed .msgpack


def occupancy_from_density(dens_morton: np.ndarray) -> np.ndarray:
    """instant-ngp's update_density_grid_mean_and_bitfield, as published: [n_cascades, 128^3] fp32
    densities in Morton order -> [n_cascades, 128^3] bool in linear order (x + 128*(y + 128*z)).

      mean   = sum(max(density, 0)) over the cells of cascade 0 / 128^3
      thresh = min(NERF_MIN_OPTICAL_THICKNESS = 0.01, mean)
      bit    = density > thresh                      (every cascade, the one threshold)
      then cascade c >= 1 ORs in the 2x2x2 max-pool of cascade c-1 over its central 64^3 cells.
    """
    n_casc = dens_morton.shape[0]
    mean = float(np.maximum(dens_morton[0], 0.0).sum(dtype=np.float64) / GRID ** 3)
    thresh = min(NERF_MIN_OPTICAL_THICKNESS, mean)
    occ = np.zeros((n_casc, GRID ** 3), bool)
    occ[:, _morton_order()] = dens_morton > thresh
    occ = occ.reshape(n_casc, GRID, GRID, GRID)                       # [c, z, y, x]
    h, q = GRID // 2, GRID // 4
    for c in range(1, n_casc):
        pooled = occ[c - 1].reshape(h, 2, h, 2, h, 2).any(axis=(1, 3, 5))
        occ[c, q:q + h, q:q + h, q:q + h] |= pooled
    return occ.reshape(n_casc, -1)


def _render_aabb(ra, aabb_scale):
    """snapshot.render_aabb ({"min": [x,y,z], "max": [x,y,z]} or six numbers) -> (lo xyz, hi xyz), or None
    when absent or equal to the model's whole box (the value a snapshot carries unless the GUI cropped it)."""
    if ra is None:
        return None
    v = (list(ra["min"]) + list(ra["max"])) if isinstance(ra, dict) else list(ra)
    if len(v) != 6:
        raise ValueError("render_aabb must hold six numbers")
    v = tuple(float(x) for x in v)
    half = 0.5 * aabb_scale
    if all(a <= 0.5 - half for a in v[:3]) and all(b >= 0.5 + half for b in v[3:]):
        return None
    return v


def load_ingp(path: str):
    """-> (NerfModel, info) where info carries training_views (intrinsics per image), dataset
    scale/offset, aabb_scale and the snapshot's background colour if present."""
    import msgpack
    cfg = msgpack.unpackb(_decompress(open(path, "rb").read()), raw=False, strict_map_key=False)
    snap = cfg["snapshot"]
    enc = cfg["encoding"]
    if enc.get("otype", "HashGrid") not in ("HashGrid", "Grid") or enc.get("type", "Hash") != "Hash":
        raise ValueError(f"unsupported position encoding {enc}")
    net, rgb = cfg["network"], cfg["rgb_network"]
    if (net.get("n_neurons", 64), rgb.get("n_neurons", 64)) != (64, 64) or net.get("n_hidden_layers", 1) != 1 \
            or rgb.get("n_hidden_layers", 2) != 2:
        raise ValueError("only the 32->64->16 density and 32->64->64->16 colour MLPs are implemented")
    nerf = snap.get("nerf", {})
    ds = nerf.get("dataset", {})
    aabb_scale = int(nerf.get("aabb_scale", ds.get("aabb_scale", 1)))
    if aabb_scale < 1 or aabb_scale & (aabb_scale - 1) or aabb_scale > 128:
        raise NotImplementedError("aabb_scale must be a power of two <= 128")
    n_casc = aabb_scale.bit_length()
    L, F = int(enc.get("n_levels", 16)), int(enc.get("n_features_per_level", 2))
    if (L, F) not in ((16, 2), (8, 4)):
        raise ValueError(f"hash grid layout L={L}, F={F}: only L=16,F=2 and L=8,F=4 (32 network inputs) are implemented")
    levels = grid_levels(L, F, int(enc.get("log2_hashmap_size", 19)), int(enc.get("base_resolution", 16)),
                         enc.get("per_level_scale"), aabb_scale)
    if snap.get("params_type", "__half") != "__half":
        raise ValueError("params_type must be __half")
    params = np.frombuffer(snap["params_binary"], np.float16)
    n_in = L * F
    sizes = [64 * n_in, 16 * 64, 64 * 32, 64 * 64, 16 * 64, levels.n_entries * F]
    if params.size != sum(sizes):
        raise ValueError(f"params_binary holds {params.size} halves, expected {sum(sizes)} for this config")
    parts = np.split(params, np.cumsum(sizes)[:-1])
    dw1, dw2 = parts[0].reshape(64, n_in), parts[1].reshape(16, 64)
    cw1, cw2, cw3 = parts[2].reshape(64, 32), parts[3].reshape(64, 64), parts[4].reshape(16, 64)
    grid = parts[5].reshape(levels.n_entries, F)
    if int(snap.get("density_grid_size", GRID)) != GRID:
        raise ValueError("density_grid_size must be 128")
    dens = np.frombuffer(snap["density_grid_binary"], np.float16).astype(np.float32)
    if dens.size != n_casc * GRID ** 3:
        raise ValueError(f"density grid must hold {n_casc} cascade(s) of 128^3 values")
    occ_lin = occupancy_from_density(dens.reshape(n_casc, -1))
    model = NerfModel(levels, grid.copy(), dw1.copy(), dw2.copy(), cw1.copy(), cw2.copy(), cw3.copy(),
                      np.packbits(occ_lin.reshape(-1).astype(np.uint8), bitorder="little"), aabb_scale,
                      _render_aabb(snap.get("render_aabb"), aabb_scale))
    views = []
    for md in ds.get("metadata", []):
        w, h = md["resolution"]
        fx, fy = md["focal_length"]
        cx, cy = md["principal_point"]
        ln = md.get("lens") or {}
        lens = tuple(float(ln[k]) for k in ("k1", "k2", "p1", "p2")) if all(k in ln for k in ("k1", "k2", "p1", "p2")) and "k3" not in ln else None
        views.append(dict(fx=float(fx), fy=float(fy), cx=float(cx) * w, cy=float(cy) * h, w=int(w), h=int(h), lens=lens))
    info = dict(training_views=views, dataset_scale=float(ds.get("scale", 1.0)),
                dataset_offset=tuple(ds.get("offset", (0.5, 0.5, 0.5))), aabb_scale=aabb_scale,
                background_color=snap.get("background_color"), training_step=snap.get("training_step"))
    return model, info
