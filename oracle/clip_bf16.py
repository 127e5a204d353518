"""TEST INFRASTRUCTURE ONLY — the "ideal bf16 tower": the fp32 restatement of oracle/clip_ref.py (reference clip_scoring.py:150,180-181,
Hugging Face CLIPModel) with EVERY matrix-product operand rounded to bf16 (round to nearest even) and everything else — accumulation,
LayerNorm, softmax, residual stream, GELU — in fp32.  north_star prescribes bf16 MFMA inputs with fp32 accumulation; this is what such an
implementation computes when nothing else loses precision, so its distance from the fp32 oracle is the FLOOR any bf16 tower has on a given
(weights, input) pair.  The parity tests use it where the fp32 distance itself says little: under adversarial (trained-like) statistics on
white-noise inputs the floor alone is ~1.8e-3 of a logit — above north_star's 1e-3 — and the HIP tower is then held to the floor instead
(tests/test_gpu_parity.py::test_vit_against_round5_hf_goldens).  The product never imports this file."""
from __future__ import annotations

import numpy as np

from oracle import clip_ref


def round_bf16(a):
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(np.float32)


def _linear(x, sd, name):
    y = round_bf16(x) @ round_bf16(sd[name + ".weight"]).T
    if name + ".bias" in sd:
        y = y + sd[name + ".bias"]
    return y.astype(np.float32)


def _attention(x, sd, pre, n_heads):
    B, T, D = x.shape
    dh = D // n_heads
    q, k, v = (round_bf16(_linear(x, sd, f"{pre}.{n}_proj")) for n in "qkv")       # q / k / v are stored as bf16
    sh = lambda t: t.reshape(B, T, n_heads, dh).transpose(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q @ k.transpose(0, 1, 3, 2)) * np.float32(dh ** -0.5)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    o = (round_bf16(p) @ v) / p.sum(-1, keepdims=True)                               # P is the second product's bf16 operand
    return _linear(o.transpose(0, 2, 1, 3).reshape(B, T, D).astype(np.float32), sd, pre + ".out_proj")


def vision_embeds(pixel_values, sd, cfg):
    """pixel_values [B,3,S,S] f32 -> L2-normalised image_embeds [B,D], bf16 operands / fp32 everything else."""
    P, d = cfg["patch_size"], cfg["hidden_size"]
    B, _, S, _ = pixel_values.shape
    g = S // P
    patches = pixel_values.reshape(B, 3, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    w = sd["vision_model.embeddings.patch_embedding.weight"].reshape(d, 3 * P * P)
    x = (round_bf16(patches) @ round_bf16(w).T).astype(np.float32)
    cls = np.broadcast_to(sd["vision_model.embeddings.class_embedding"], (B, 1, d))
    x = np.concatenate([cls, x], 1) + sd["vision_model.embeddings.position_embedding.weight"][None]
    x = clip_ref.layer_norm(x, sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"]).astype(np.float32)
    for l in range(cfg["num_layers"]):
        p = f"vision_model.encoder.layers.{l}"
        h = clip_ref.layer_norm(x, sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"])
        x = x + _attention(h, sd, p + ".self_attn", cfg["num_heads"])
        h = clip_ref.layer_norm(x, sd[p + ".layer_norm2.weight"], sd[p + ".layer_norm2.bias"])
        x = x + _linear(clip_ref.quick_gelu(_linear(h, sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
    pooled = clip_ref.layer_norm(x[:, 0], sd["vision_model.post_layernorm.weight"], sd["vision_model.post_layernorm.bias"])
    e = (pooled @ sd["visual_projection.weight"].T).astype(np.float32)
    return e / np.linalg.norm(e, axis=-1, keepdims=True)
