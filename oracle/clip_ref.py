"""numpy fp32 restatement of the CLIP forward the reference runs at clip_scoring.py:150-181
(Hugging Face transformers CLIPModel, pinned 4.27.3 in requirements.txt:256; semantics
checked against the installed 5.x sources — SURVEY.md Appendix B).

TEST INFRASTRUCTURE ONLY — see oracle/d2r_oracle.c.  The arithmetic lives in a third-party
dependency, so this file is pinned by golden vectors produced here from the real
`transformers.CLIPModel` with seeded random weights (tests/golden/make_goldens.py);
pretrained-weight parity is unpinned because no checkpoint is available offline.

Weights are a dict keyed by the Hugging Face state_dict names.
"""
from __future__ import annotations

import numpy as np


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps))) * w + b


def quick_gelu(x):
    return x * (1.0 / (1.0 + np.exp(-1.702 * x))).astype(np.float32)


def _linear(x, sd, name):
    y = x @ sd[name + ".weight"].T
    if name + ".bias" in sd:
        y = y + sd[name + ".bias"]
    return y.astype(np.float32)


def _attention(x, sd, pre, n_heads, causal):
    B, T, D = x.shape
    dh = D // n_heads
    q = _linear(x, sd, pre + ".q_proj") * np.float32(dh ** -0.5)
    k = _linear(x, sd, pre + ".k_proj")
    v = _linear(x, sd, pre + ".v_proj")
    sh = lambda t: t.reshape(B, T, n_heads, dh).transpose(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = q @ k.transpose(0, 1, 3, 2)
    if causal:
        s = s + np.triu(np.full((T, T), -np.inf, np.float32), 1)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
    return _linear(o.astype(np.float32), sd, pre + ".out_proj")


def _encoder(x, sd, pre, n_layers, n_heads, causal, hidden_out=None):
    for l in range(n_layers):
        p = f"{pre}.layers.{l}"
        h = layer_norm(x, sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"])
        x = x + _attention(h, sd, p + ".self_attn", n_heads, causal)
        h = layer_norm(x, sd[p + ".layer_norm2.weight"], sd[p + ".layer_norm2.bias"])
        h = quick_gelu(_linear(h, sd, p + ".mlp.fc1"))
        x = x + _linear(h, sd, p + ".mlp.fc2")
        if hidden_out is not None:
            hidden_out.append(x.copy())
    return x


def vision_embeds(pixel_values, sd, cfg, hidden_out=None):
    """pixel_values [B,3,S,S] f32 -> L2-normalised image_embeds [B,D]."""
    P, d = cfg["patch_size"], cfg["hidden_size"]
    B, _, S, _ = pixel_values.shape
    g = S // P
    # conv2d kernel=stride=P, no bias == per-patch GEMM with the flattened [d, 3*P*P] kernel
    patches = pixel_values.reshape(B, 3, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    w = sd["vision_model.embeddings.patch_embedding.weight"].reshape(d, 3 * P * P)
    x = (patches @ w.T).astype(np.float32)
    cls = np.broadcast_to(sd["vision_model.embeddings.class_embedding"], (B, 1, d))
    x = np.concatenate([cls, x], 1) + sd["vision_model.embeddings.position_embedding.weight"][None]
    x = layer_norm(x, sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"])
    if hidden_out is not None:
        hidden_out.append(x.copy())
    x = _encoder(x.astype(np.float32), sd, "vision_model.encoder", cfg["num_layers"], cfg["num_heads"],
                 False, hidden_out)
    pooled = layer_norm(x[:, 0], sd["vision_model.post_layernorm.weight"], sd["vision_model.post_layernorm.bias"])
    e = (pooled @ sd["visual_projection.weight"].T).astype(np.float32)
    return e / np.linalg.norm(e, axis=-1, keepdims=True)


def text_embeds(input_ids, sd, cfg):
    """input_ids [C,T] (EOS id = largest id, pooled at argmax) -> L2-normalised [C,D]."""
    ids = np.asarray(input_ids)
    Cn, T = ids.shape
    x = sd["text_model.embeddings.token_embedding.weight"][ids] + \
        sd["text_model.embeddings.position_embedding.weight"][None, :T]
    x = _encoder(x.astype(np.float32), sd, "text_model.encoder", cfg["text_layers"], cfg["text_heads"], True)
    x = layer_norm(x, sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"])
    pooled = x[np.arange(Cn), ids.argmax(-1)]
    e = (pooled @ sd["text_projection.weight"].T).astype(np.float32)
    return e / np.linalg.norm(e, axis=-1, keepdims=True)


def logits_per_image(img_e, txt_e, logit_scale):
    """exp(logit_scale) * I @ T^T  (HF CLIPModel.forward)."""
    return (np.float32(np.exp(np.float32(logit_scale))) * (img_e @ txt_e.T)).astype(np.float32)
