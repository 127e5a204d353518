#!/usr/bin/env python3
"""Run on the GPU box with libd2r built with EXTRA=-DD2R_ATTN_STAMPS: average shader-clock cycles wave 0 of a
k_attention workgroup spends issuing its loads / V transposes, waiting for K, at the barrier, and computing."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import engine, _lib
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict

ctx = engine.Context(0)
cfg = CLIP_CONFIGS["vit_b16"]
sc = engine.ClipScorer(ctx, cfg, random_clip_state_dict(cfg, seed=6, text=False))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pv = np.random.default_rng(0).standard_normal((n, 3, 224, 224), dtype=np.float32)
lib = _lib.load()
out = (C.c_ulonglong * 8)()
sc.embed_pixels(pv)
lib.d2r_debug_attn_stamps(out, 1)
sc.embed_pixels(pv)
lib.d2r_debug_attn_stamps(out, 0)
t = out[4]
print(f"workgroups {t}: issue loads+V transposes {out[0] / t:.0f}, wait K {out[1] / t:.0f}, barrier {out[2] / t:.0f}, compute+store {out[3] / t:.0f} cycles")
