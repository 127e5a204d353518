"""Test-fixture writer of `.ingp` snapshots in the layout dream2real_amd.ingp.load_ingp reads
(SURVEY.md Appendix A).  Not product code."""
from __future__ import annotations

import zlib

import numpy as np

from dream2real_amd.ingp import _morton_order
from dream2real_amd.scene import GRID, NerfModel


def save_ingp(path: str, model: NerfModel, training_views=None, dataset_scale: float = 1.0,
              dataset_offset=(0.0, 0.3, 0.5), density_value: float = 1.0, background_color=None):
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
            **({"background_color": list(background_color)} if background_color is not None else {}),
            **({"render_aabb": {"min": list(model.render_aabb[:3]), "max": list(model.render_aabb[3:])}}
               if getattr(model, "render_aabb", None) is not None else {}),
            "nerf": {"aabb_scale": int(getattr(model, "aabb_scale", 1)), "dataset": {
                "n_images": len(views), "scale": dataset_scale, "offset": list(dataset_offset),
                "aabb_scale": int(getattr(model, "aabb_scale", 1)),
                "metadata": [{"resolution": [v["w"], v["h"]], "focal_length": [v["fx"], v["fy"]],
                              "principal_point": [v["cx"] / v["w"], v["cy"] / v["h"]],
                              "lens": (dict(zip(("k1", "k2", "p1", "p2"), [float(x) for x in v["lens"]])) if v.get("lens") is not None else {})}
                             for v in views]}},
        },
    }
    with open(path, "wb") as f:
        f.write(zlib.compress(msgpack.packb(cfg, use_bin_type=True), 1))
