"""Numpy emulation of an e4m3 ViT-B/16 (MX block-32 or per-row scales on activations and weights) against the fp32
oracle: the measurement behind "no fp8 tower" in DESIGN.md section 7.  Test-side diagnostic (uses oracle/)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import clip_ref
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
cfg = CLIP_CONFIGS["vit_b16"]; sd = random_clip_state_dict(cfg, seed=6, text=False)
def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7fff
    return ((u + r) & 0xffff0000).view(np.float32)
def e4m3(x):
    # round to nearest e4m3 (3 mantissa bits, min normal 2^-6, max 448), x already scaled
    x = np.clip(x, -448, 448).astype(np.float32)
    a = np.abs(x); e = np.floor(np.log2(np.maximum(a, 2.0**-9))); e = np.maximum(e, -6)
    q = np.round(a / 2.0**(e-3)) * 2.0**(e-3)
    return np.sign(x) * q
def mx_quant(x, block=32):
    # per-row blocks of `block` along last axis, E8M0 scale = 2^ceil(log2(amax/448))
    sh = x.shape; xb = x.reshape(-1, sh[-1] // block, block)
    amax = np.abs(xb).max(-1, keepdims=True)
    s = 2.0 ** np.ceil(np.log2(np.maximum(amax, 1e-30) / 448.0))
    return (e4m3(xb / s) * s).reshape(sh).astype(np.float32)
def row_quant(x):
    amax = np.abs(x).max(-1, keepdims=True); s = np.maximum(amax, 1e-30) / 448.0
    return (e4m3(x / s) * s).astype(np.float32)
r = np.random.Generator(np.random.PCG64(3))
pv = r.standard_normal((3, 3, 224, 224), dtype=np.float32)
ref = clip_ref.vision_embeds(pv, sd, cfg)
def run(qa, qw):
    sdq = dict(sd)
    for k, v in sd.items():
        if v.ndim == 2 and 'position' not in k and 'projection' not in k and 'encoder' in k:
            sdq[k] = qw(v)
    def _linear(x, sd_, name):
        xin = qa(x) if 'encoder' in name else x
        y = xin @ sd_[name + ".weight"].T
        if name + ".bias" in sd_: y = y + sd_[name + ".bias"]
        return y.astype(np.float32)
    old = clip_ref._linear; clip_ref._linear = _linear
    try: e = clip_ref.vision_embeds(pv, sdq, cfg)
    finally: clip_ref._linear = old
    return (1 - (e * ref).sum(-1)).max(), np.abs(e - ref).max()
print('bf16      ', run(bf16, bf16))
print('mx-fp8 b32', run(mx_quant, mx_quant))
print('row-fp8   ', run(row_quant, row_quant))
