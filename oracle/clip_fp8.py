"""numpy restatement of the vision tower's fp8 mode (library option "vit_fp8"; BASELINE.json configs[4] names an
"fp8 MFMA ViT").  The reference has no fp8 path (clip_scoring.py:150-181 runs CLIPModel in fp32/fp16): this file is the
SPECIFICATION of the quantisation the HIP kernels implement (dream2real_amd/csrc/clip.hip, "fp8 blocks"), stated on top of the
fp32 restatement oracle/clip_ref.py, so that the tests can separate two questions:
  * does the HIP path implement this specification?   (HIP vs this file: the bf16 path's bar)
  * what does the specification cost against fp32?     (this file / HIP vs oracle/clip_ref.py: measured and reported)

TEST INFRASTRUCTURE ONLY — see oracle/d2r_oracle.c.

Format (OCP "MX"-style, 8-bit floats e4m3fn: 4 exponent bits, bias 7, 3 mantissa bits, largest finite value 448, no infinity):
  activations  one E8M0 scale byte per (row, group of 64 consecutive columns): byte = clamp(e - 8, 1, 253) with e the biased
               exponent of the group's largest magnitude, one more when that magnitude's mantissa is >= 1.75 (so that
               amax / 2^(byte - 127) <= 448); element = round-to-nearest-even e4m3 of x / 2^(byte - 127)
  weights      the bf16-rounded weight (what the bf16 path multiplies with), ONE such scale per matrix
  product      exact products, fp32 accumulation; times the matrix scale, plus the bias
Quantised: the inputs of q/k/v_proj (LayerNorm 1 output), out_proj (attention output: a group = a head), fc1 (LayerNorm 2
output) and fc2 (quick_gelu output) of the layers in `layers`.
"""
from __future__ import annotations

import numpy as np

from . import clip_ref


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def scale_byte(amax):
    """E8M0 byte of a group whose largest magnitude is amax (fp32 array)."""
    bits = np.ascontiguousarray(amax, np.float32).view(np.uint32).astype(np.int64)
    e = (bits + 0x200000) >> 23
    return np.clip(e - 8, 1, 253).astype(np.int32)


def e4m3_round(x):
    """round-to-nearest-even onto the e4m3fn grid; |x| <= 448 is the caller's business (the scales guarantee it)."""
    x = np.asarray(x, np.float32)
    a = np.abs(x).astype(np.float64)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                       # subnormals share the exponent of the smallest normal
    step = 2.0 ** (e - 3.0)
    q = np.round(a / step) * step                 # numpy rounds halves to even
    return (np.sign(x) * q).astype(np.float32)


def e4m3_encode(q):
    """byte pattern of values already on the grid."""
    q = np.asarray(q, np.float32)
    a = np.abs(q).astype(np.float64)
    s = (np.signbit(q)).astype(np.uint8) << 7
    sub = a < 2.0 ** -6
    e = np.floor(np.log2(np.where(sub | (a == 0), 1.0, a)))
    mant = np.where(sub, a / 2.0 ** -9, (a / 2.0 ** e - 1.0) * 8.0)
    expo = np.where(sub, 0, e + 7)
    return (s | (expo.astype(np.uint8) << 3) | np.round(mant).astype(np.uint8)).astype(np.uint8)


def quant_act(x, group=64):
    """x [..., K] -> (dequantised values, e4m3 values before scaling, scale bytes [..., K/group])."""
    x = np.asarray(x, np.float32)
    sh = x.shape
    g = x.reshape(sh[:-1] + (sh[-1] // group, group))
    sb = scale_byte(np.abs(g).max(-1))
    inv = np.exp2((127 - sb).astype(np.float32))[..., None]
    q = e4m3_round(g * inv)
    deq = q * np.exp2((sb - 127).astype(np.float32))[..., None]
    return deq.reshape(sh).astype(np.float32), q.reshape(sh), sb


def quant_weight(w):
    """w [N, K] fp32 -> (dequantised, scale) of the bf16-rounded matrix with one power-of-two scale."""
    wb = bf16_round(w)
    sb = int(scale_byte(np.array([np.abs(wb).max()], np.float32))[0])
    scale = np.float32(2.0 ** (sb - 127))
    return (e4m3_round(wb / scale) * scale).astype(np.float32), scale


def linear_fp8(x, w, b):
    xq, _, _ = quant_act(x)
    wq, _ = quant_weight(w)
    return (xq.astype(np.float64) @ wq.astype(np.float64).T + b).astype(np.float32)


def fp8_layers(cfg, l0_reuse: bool, cls_last: bool = True):
    """the layers the library runs in fp8 (d2r_clip_forward): not the first when its rows are reused, not the last."""
    n = cfg["num_layers"]
    return list(range(1 if l0_reuse else 0, n - 1 if cls_last else n))


def vision_embeds(pixel_values, sd, cfg, layers):
    """clip_ref.vision_embeds with the Linear inputs of `layers` quantised as above."""
    cache = {}

    def _linear(x, sd_, name):
        parts = name.split(".")
        if "layers" in parts and int(parts[parts.index("layers") + 1]) in layers and name.startswith("vision_model.encoder"):
            if name not in cache:
                cache[name] = quant_weight(sd_[name + ".weight"])[0]
            xq = quant_act(x)[0]
            return (xq @ cache[name].T + sd_[name + ".bias"]).astype(np.float32)
        return old(x, sd_, name)

    old = clip_ref._linear
    clip_ref._linear = _linear
    try:
        return clip_ref.vision_embeds(pixel_values, sd, cfg)
    finally:
        clip_ref._linear = old
