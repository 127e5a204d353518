"""`get_vis_ngps` (reference reconstruction/ngp_visual_model.py:20-29): the visual models of the task — the
foreground (movable object) and background NeRFs — as Testbeds.

Only the cached-snapshot half is part of the path: `use_cache=True` loads
`<data_dir>/fg_base.ingp` or `bg_base.ingp` through the library's snapshot reader (d2r_nerf_load_ingp).  Training the
models (`use_cache=False`: image export + `build_vis_model`, reference :30-60) is NeRF training and stays out of scope;
it raises with a message instead of silently returning something else.
"""
from __future__ import annotations

import os


def get_vis_ngps(rgbs, movable_masks, scene_type, use_cache=False, data_dir=None, fg=True, render_distract=False, *, ctx=None):
    """Same positional arguments as the reference; `ctx` (an engine.Context) stands where the reference constructs
    `ngp.Testbed(ngp.TestbedMode.Nerf)` on the current CUDA device."""
    from .engine import Testbed
    if not use_cache:
        raise NotImplementedError("get_vis_ngps(use_cache=False) trains the NeRFs (reference ngp_visual_model.py:30-60): NeRF "
                                  "training is outside the render-and-score path; train with instant-ngp and pass use_cache=True")
    if ctx is None:
        raise ValueError("get_vis_ngps needs ctx=engine.Context(device)")
    if data_dir is None:
        raise ValueError("get_vis_ngps(use_cache=True) needs data_dir")
    print("Using cached fg model for movable object")          # the reference prints this for both models (:22)
    return Testbed.from_snapshot(ctx, os.path.join(data_dir, "fg_base.ingp" if fg else "bg_base.ingp"))
