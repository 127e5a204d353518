#!/usr/bin/env python3
"""Run on the GPU box with libd2r built with DEV=1 EXTRA=-DD2R_GEMM_STAMPS (tools/gemm_stamps.py [images] [model] [fp8]): average shader-clock cycles
per tile that wave 0 of a k_gemm8 workgroup spends waiting for the drained queue, in the K loop and
in the epilogue, per epilogue kind, over one bench-sized ViT forward."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import engine, _lib
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict

ctx = engine.Context(0)
cfg = CLIP_CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "vit_b16"]
if len(sys.argv) > 3 and sys.argv[3] == "fp8":
    ctx.set_option("vit_fp8", 1)
sc = engine.ClipScorer(ctx, cfg, random_clip_state_dict(cfg, seed=6, text=False))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pv = np.random.default_rng(0).standard_normal((n, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
lib = _lib.load()
KINDS = 13
out = (C.c_ulonglong * (4 * KINDS))()
sc.embed_pixels(pv)
lib.d2r_debug_gemm_stamps(out, 1)
sc.embed_pixels(pv)
lib.d2r_debug_gemm_stamps(out, 0)
names = ["patch (fp32)", "bias bf16", "bias gelu bf16", "bias + fp32 residual", "QKV (LN-folded, bf16)", "fc1 (LN-folded, gelu, bf16)",
         "resid + stats, fp32 + copy", "resid + stats, bf16", "resid + stats, split bf16", "out-proj + fc2 (resid + stats, hi + lo byte)",
         "fp8: QKV (bf16 out)", "fp8: fc1 (gelu, e4m3 out)", "fp8: out-proj + fc2 (resid + stats)"]
for e in range(KINDS):
    w, k, ep, t = out[4 * e:4 * e + 4]
    if t:
        print(f"{names[e]:44s} tiles/WG-wave0 {t:7d}  drain {w / t:9.0f}  K loop {k / t:9.0f}  epilogue {ep / t:9.0f} cycles/tile")
