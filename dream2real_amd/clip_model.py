"""CLIP model descriptions and weight plumbing for the scoring half of the path.

The reference scores renders with Hugging Face `CLIPModel`
("openai/clip-vit-large-patch14-336", reference clip_scoring.py:150-151).  Here a
checkpoint is a plain dict keyed by the Hugging Face state_dict names; the vision tower
is flattened into one fp32 blob in the order `include/d2r.h` documents for
`d2r_clip_create`.  No pretrained weights are available offline, so tests and bench.py use
seeded random weights of the exact architecture (`random_clip_state_dict`).
"""
from __future__ import annotations

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

CLIP_CONFIGS = {
    # BASELINE.json configs 1-3 (ViT-B/16, 197 tokens)
    "vit_b16": dict(patch_size=16, hidden_size=768, num_layers=12, num_heads=12, mlp=3072, image_size=224,
                    proj=512, text_hidden=512, text_layers=12, text_heads=8, text_mlp=2048,
                    vocab=49408, ctx=77),
    # BASELINE.json config 4 (ViT-L/14, 257 tokens)
    "vit_l14": dict(patch_size=14, hidden_size=1024, num_layers=24, num_heads=16, mlp=4096, image_size=224,
                    proj=768, text_hidden=768, text_layers=12, text_heads=12, text_mlp=3072,
                    vocab=49408, ctx=77),
    # the reference's own model (clip_scoring.py:150), 577 tokens
    "vit_l14_336": dict(patch_size=14, hidden_size=1024, num_layers=24, num_heads=16, mlp=4096, image_size=336,
                        proj=768, text_hidden=768, text_layers=12, text_heads=12, text_mlp=3072,
                        vocab=49408, ctx=77),
    # shallow ViT-L geometries (patch 14 -> padded patch GEMM, 257 / 577 tokens, d=1024) for parity tests
    "vit_l14_x2": dict(patch_size=14, hidden_size=1024, num_layers=2, num_heads=16, mlp=4096, image_size=224,
                       proj=768, text_hidden=768, text_layers=1, text_heads=12, text_mlp=3072,
                       vocab=49408, ctx=77),
    "vit_l14_336_x1": dict(patch_size=14, hidden_size=1024, num_layers=1, num_heads=16, mlp=4096, image_size=336,
                           proj=768, text_hidden=768, text_layers=1, text_heads=12, text_mlp=3072,
                           vocab=49408, ctx=77),
    # 2-layer model with the same code paths, for unit tests and committed goldens
    "vit_tiny": dict(patch_size=16, hidden_size=128, num_layers=2, num_heads=2, mlp=256, image_size=64,
                     proj=64, text_hidden=128, text_layers=2, text_heads=2, text_mlp=256,
                     vocab=512, ctx=16),
}


def n_tokens(cfg) -> int:
    return (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1


def random_clip_state_dict(cfg, seed: int = 6, text: bool = True, logit_scale: float = 4.6052) -> dict:
    """Seeded (numpy PCG64) weights with fan-in scaling so the network is input-sensitive.
    Identical on every machine, independent of torch's RNG."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def normal(shape, std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)

    def lin(name, n_out, n_in, bias=True, gain=1.0):
        sd[name + ".weight"] = normal((n_out, n_in), gain / np.sqrt(n_in))
        if bias:
            sd[name + ".bias"] = normal((n_out,), 0.02)

    def ln(name, d):
        sd[name + ".weight"] = (1.0 + normal((d,), 0.05)).astype(np.float32)
        sd[name + ".bias"] = normal((d,), 0.02)

    def tower(pre, d, layers, mlp):
        for l in range(layers):
            p = f"{pre}.layers.{l}"
            ln(p + ".layer_norm1", d)
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                lin(f"{p}.self_attn.{nm}", d, d)
            ln(p + ".layer_norm2", d)
            lin(p + ".mlp.fc1", mlp, d)
            lin(p + ".mlp.fc2", d, mlp)

    d, P = cfg["hidden_size"], cfg["patch_size"]
    sd["vision_model.embeddings.patch_embedding.weight"] = normal((d, 3, P, P), 1.0 / np.sqrt(3 * P * P))
    sd["vision_model.embeddings.class_embedding"] = normal((d,), 0.5)
    sd["vision_model.embeddings.position_embedding.weight"] = normal((n_tokens(cfg), d), 0.1)
    ln("vision_model.pre_layrnorm", d)
    tower("vision_model.encoder", d, cfg["num_layers"], cfg["mlp"])
    ln("vision_model.post_layernorm", d)
    lin("visual_projection", cfg["proj"], d, bias=False)
    if text:
        td = cfg["text_hidden"]
        sd["text_model.embeddings.token_embedding.weight"] = normal((cfg["vocab"], td), 0.5)
        sd["text_model.embeddings.position_embedding.weight"] = normal((cfg["ctx"], td), 0.1)
        tower("text_model.encoder", td, cfg["text_layers"], cfg["text_mlp"])
        ln("text_model.final_layer_norm", td)
        lin("text_projection", cfg["proj"], td, bias=False)
    sd["logit_scale"] = np.float32(logit_scale)
    return sd


def vision_blob_order(cfg):
    """(name, shape) of every tensor in the `d2r_clip_create` weight blob, in order."""
    d, P, mlp = cfg["hidden_size"], cfg["patch_size"], cfg["mlp"]
    out = [("vision_model.embeddings.patch_embedding.weight", (d, 3 * P * P)),
           ("vision_model.embeddings.class_embedding", (d,)),
           ("vision_model.embeddings.position_embedding.weight", (n_tokens(cfg), d)),
           ("vision_model.pre_layrnorm.weight", (d,)), ("vision_model.pre_layrnorm.bias", (d,))]
    for l in range(cfg["num_layers"]):
        p = f"vision_model.encoder.layers.{l}"
        out += [(p + ".layer_norm1.weight", (d,)), (p + ".layer_norm1.bias", (d,))]
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(f"{p}.self_attn.{nm}.weight", (d, d)), (f"{p}.self_attn.{nm}.bias", (d,))]
        out += [(p + ".layer_norm2.weight", (d,)), (p + ".layer_norm2.bias", (d,)),
                (p + ".mlp.fc1.weight", (mlp, d)), (p + ".mlp.fc1.bias", (mlp,)),
                (p + ".mlp.fc2.weight", (d, mlp)), (p + ".mlp.fc2.bias", (d,))]
    out += [("vision_model.post_layernorm.weight", (d,)), ("vision_model.post_layernorm.bias", (d,)),
            ("visual_projection.weight", (cfg["proj"], d))]
    return out


def pack_vision_weights(sd: dict, cfg) -> np.ndarray:
    """Flatten the vision tower + projection into the fp32 blob `d2r_clip_create` takes."""
    parts = []
    for name, shape in vision_blob_order(cfg):
        a = np.asarray(sd[name], np.float32).reshape(shape)
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def text_blob_order(cfg):
    """(name, shape) of every tensor in the `d2r_text_create` weight blob, in order."""
    d, mlp = cfg["text_hidden"], cfg["text_mlp"]
    out = [("text_model.embeddings.token_embedding.weight", (cfg["vocab"], d)),
           ("text_model.embeddings.position_embedding.weight", (cfg["ctx"], d))]
    for l in range(cfg["text_layers"]):
        p = f"text_model.encoder.layers.{l}"
        out += [(p + ".layer_norm1.weight", (d,)), (p + ".layer_norm1.bias", (d,))]
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(f"{p}.self_attn.{nm}.weight", (d, d)), (f"{p}.self_attn.{nm}.bias", (d,))]
        out += [(p + ".layer_norm2.weight", (d,)), (p + ".layer_norm2.bias", (d,)),
                (p + ".mlp.fc1.weight", (mlp, d)), (p + ".mlp.fc1.bias", (mlp,)),
                (p + ".mlp.fc2.weight", (d, mlp)), (p + ".mlp.fc2.bias", (d,))]
    out += [("text_model.final_layer_norm.weight", (d,)), ("text_model.final_layer_norm.bias", (d,)),
            ("text_projection.weight", (cfg["proj"], d))]
    return out


def pack_text_weights(sd: dict, cfg) -> np.ndarray:
    return np.ascontiguousarray(np.concatenate(
        [np.asarray(sd[n], np.float32).reshape(shape).reshape(-1) for n, shape in text_blob_order(cfg)]))


def infer_clip_config(sd: dict) -> dict:
    """CLIP geometry from the tensor shapes of a Hugging Face `CLIPModel` state dict (head_dim 64, as
    in every OpenAI CLIP ViT)."""
    pe = sd["vision_model.embeddings.patch_embedding.weight"]
    d, P = int(pe.shape[0]), int(pe.shape[-1])
    n_tok = int(sd["vision_model.embeddings.position_embedding.weight"].shape[0])
    g = int(round((n_tok - 1) ** 0.5))
    if g * g + 1 != n_tok:
        raise ValueError(f"{n_tok} vision positions is not a square grid plus the class token")

    def n_layers(prefix):
        n = 0
        while f"{prefix}.layers.{n}.layer_norm1.weight" in sd:
            n += 1
        return n

    cfg = dict(patch_size=P, hidden_size=d, num_layers=n_layers("vision_model.encoder"), num_heads=d // 64,
               mlp=int(sd["vision_model.encoder.layers.0.mlp.fc1.weight"].shape[0]), image_size=g * P,
               proj=int(sd["visual_projection.weight"].shape[0]))
    if "text_model.embeddings.token_embedding.weight" in sd:
        te = sd["text_model.embeddings.token_embedding.weight"]
        td = int(te.shape[1])
        cfg.update(text_hidden=td, text_layers=n_layers("text_model.encoder"), text_heads=td // 64,
                   text_mlp=int(sd["text_model.encoder.layers.0.mlp.fc1.weight"].shape[0]),
                   vocab=int(te.shape[0]), ctx=int(sd["text_model.embeddings.position_embedding.weight"].shape[0]))
    return cfg


def load_clip_safetensors(path: str):
    """-> (cfg, state_dict of fp32 numpy arrays) from a Hugging Face CLIP `model.safetensors`
    (the reference's `CLIPModel.from_pretrained(...)`, clip_scoring.py:150; fp32, fp16 or bf16 on
    disk).  Buffers that are not weights (`position_ids`) are dropped."""
    from safetensors.torch import load_file
    raw = load_file(path)
    sd = {k: v.float().numpy() for k, v in raw.items() if not k.endswith("position_ids")}
    if "logit_scale" in sd:
        sd["logit_scale"] = np.float32(sd["logit_scale"].reshape(-1)[0])
    return infer_clip_config(sd), sd
