"""Reader (and test writer) for instant-ngp `.ingp` NeRF snapshots.

The reference loads `fg_base.ingp` / `bg_base.ingp` through `pyngp.Testbed.load_snapshot`
(reference reconstruction/ngp_visual_model.py:24-28).  No snapshot and no instant-ngp source is
available offline, so this module is written against the format as believed (SURVEY.md §3.4 and
Appendix A) and is exercised on snapshots produced by `save_ingp` below in the same layout.
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
    if aabb_scale not in (1, 2):
        raise NotImplementedError("aabb_scale > 2 (more than two occupancy cascades) is not implemented")
    n_casc = aabb_scale.bit_length()
    L, F = int(enc.get("n_levels", 16)), int(enc.get("n_features_per_level", 2))
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
    pos = dens[dens > 0]
    thresh = min(float(pos.mean()) if pos.size else 0.0, NERF_MIN_OPTICAL_THICKNESS)
    occ_lin = np.zeros((n_casc, GRID ** 3), bool)
    occ_lin[:, _morton_order()] = dens.reshape(n_casc, -1) > thresh           # Morton order within a cascade
    model = NerfModel(levels, grid.copy(), dw1.copy(), dw2.copy(), cw1.copy(), cw2.copy(), cw3.copy(),
                      np.packbits(occ_lin.reshape(-1).astype(np.uint8), bitorder="little"), aabb_scale)
    views = []
    for md in ds.get("metadata", []):
        w, h = md["resolution"]
        fx, fy = md["focal_length"]
        cx, cy = md["principal_point"]
        views.append(dict(fx=float(fx), fy=float(fy), cx=float(cx) * w, cy=float(cy) * h, w=int(w), h=int(h)))
    info = dict(training_views=views, dataset_scale=float(ds.get("scale", 1.0)),
                dataset_offset=tuple(ds.get("offset", (0.5, 0.5, 0.5))), aabb_scale=aabb_scale,
                background_color=snap.get("background_color"), training_step=snap.get("training_step"))
    return model, info


def save_ingp(path: str, model: NerfModel, training_views=None, dataset_scale: float = 1.0,
              dataset_offset=(0.0, 0.3, 0.5), density_value: float = 1.0):
    """Write `model` in the layout load_ingp reads (test fixture writer; occupied cells get
    `density_value`, the rest 0)."""
    import msgpack
    lv = model.levels
    params = np.concatenate([np.asarray(a, np.float16).reshape(-1) for a in
                             (model.dw1, model.dw2, model.cw1, model.cw2, model.cw3, model.grid)])
    n_casc = int(getattr(model, "aabb_scale", 1)).bit_length()
    occ_lin = np.unpackbits(model.occ_bits, bitorder="little").astype(bool).reshape(n_casc, -1)
    dens = np.where(occ_lin[:, _morton_order()], density_value, 0.0).astype(np.float16).reshape(-1)
    views = training_views or [dict(fx=924.66912, fy=926.49735, cx=654.51953, cy=355.18523, w=1280, h=720)]
    cfg = {
        "encoding": {"otype": "HashGrid", "n_levels": lv.n_levels, "n_features_per_level": lv.n_features,
                     "log2_hashmap_size": lv.log2_hashmap_size, "base_resolution": lv.base_resolution},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                    "n_neurons": 64, "n_hidden_layers": 1},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                        "n_neurons": 64, "n_hidden_layers": 2},
        "dir_encoding": {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}]},
        "snapshot": {
            "version": 1, "mode": "nerf", "n_params": int(params.size), "params_type": "__half",
            "params_binary": params.tobytes(), "density_grid_size": GRID, "density_grid_binary": dens.tobytes(),
            "nerf": {"aabb_scale": int(getattr(model, "aabb_scale", 1)), "dataset": {
                "n_images": len(views), "scale": dataset_scale, "offset": list(dataset_offset),
                "aabb_scale": int(getattr(model, "aabb_scale", 1)),
                "metadata": [{"resolution": [v["w"], v["h"]], "focal_length": [v["fx"], v["fy"]],
                              "principal_point": [v["cx"] / v["w"], v["cy"] / v["h"]]} for v in views]}},
        },
    }
    with open(path, "wb") as f:
        f.write(zlib.compress(msgpack.packb(cfg, use_bin_type=True), 1))
