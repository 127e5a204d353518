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


def adversarial_clip_state_dict(cfg, seed: int = 6, text: bool = False, *, outlier_gain: float = 150.0, n_outlier: int = 4,
                                n_marked_patches: int = 4, broad_outlier: float = 12.0, gamma_sigma: float = 0.45,
                                gamma_spikes: int = 6, common_mode: float = 2.0, sharp_fraction: float = 0.25,
                                sharp_gain: float = 2.5, info: dict | None = None) -> dict:
    """`random_clip_state_dict` plus the statistics TRAINED CLIP vision towers are known to carry and Gaussian weights do
    not — the regime the bf16 / LayerNorm-fold schedules of the library have to survive:

    * massive activations: `n_outlier` residual channels that reach ~`outlier_gain` x the typical magnitude on the class
      token and `n_marked_patches` patch tokens ("register"-like) from the middle layer on.  Built the way such channels
      arise: the position embedding of those tokens carries a marker direction, a few fc1 neurons of the middle block
      fire on it (and on nothing else), and their fc2 columns write the large value into the outlier channels, where the
      residual stream keeps it to the end; `broad_outlier`: the same channels also carry a moderate offset on EVERY token
      (fc2 bias of that block), as the outlier dimensions of trained transformers do;
    * LayerNorm gains with a heavy tail: log-normal (sigma `gamma_sigma`) with `gamma_spikes` entries per LayerNorm at 5-10x;
    * a per-token common-mode offset of the residual stream (the all-ones direction, which no LayerNorm sees but every
      folded product carries): out_proj / fc2 get a rank-one component 1 v^T sized so that the offset they add per block
      has a spread of `common_mode` / sqrt(layers) row sigmas, plus the pre-LayerNorm bias — |mean| / sigma of a token
      row ends up around `common_mode`;
    * sharp attention: the q / k rows of a fraction `sharp_fraction` of the heads scaled by `sharp_gain` (logit spread
      x gain^2), near one-hot softmax rows.
    `info` (optional dict) receives the channel / token / head choices.  Deterministic (PCG64)."""
    sd = random_clip_state_dict(cfg, seed, text=text)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    d, L, H, mlp, T = cfg["hidden_size"], cfg["num_layers"], cfg["num_heads"], cfg["mlp"], n_tokens(cfg)
    dh = d // H
    enc = "vision_model.encoder.layers."
    f32 = np.float32

    # ---- heavy-tailed LayerNorm gains (every LayerNorm of the vision tower)
    ln_names = ["vision_model.pre_layrnorm", "vision_model.post_layernorm"] + \
               [f"{enc}{l}.layer_norm{k}" for l in range(L) for k in (1, 2)]
    for nm in ln_names:
        g = np.exp(rng.standard_normal(d) * gamma_sigma).astype(f32)
        spikes = rng.choice(d, size=min(gamma_spikes, d), replace=False)
        g[spikes] = rng.uniform(5.0, 10.0, size=len(spikes)).astype(f32)
        g *= rng.choice([-1.0, 1.0], size=d, p=[0.05, 0.95]).astype(f32)          # a few negative gains, as trained models have
        sd[nm + ".weight"] = g
        sd[nm + ".bias"] = (rng.standard_normal(d) * 0.1).astype(f32)

    # ---- sharp heads
    n_sharp = int(round(sharp_fraction * H))
    sharp = {}
    for l in range(L):
        heads = rng.choice(H, size=n_sharp, replace=False)
        sharp[l] = sorted(int(h) for h in heads)
        for h in heads:
            for nm in ("q_proj", "k_proj"):
                sd[f"{enc}{l}.self_attn.{nm}.weight"][h * dh:(h + 1) * dh] *= f32(sharp_gain)

    # ---- common mode: rank-one 1 v^T in the residual writers, and a mean in the embedding LayerNorm's bias
    ones = np.ones(d, f32)
    if common_mode > 0:
        per_block = common_mode / np.sqrt(2.0 * L)              # 2 L writers whose offsets add up like a random walk
        for l in range(L):
            for nm, n_in in ((f"{enc}{l}.self_attn.out_proj", d), (f"{enc}{l}.mlp.fc2", mlp)):
                W = sd[nm + ".weight"]
                # the writer's input has rms ~ r_in per element (attention output ~ 1, GELU hidden ~ 0.4): v^T x has spread |v| r_in
                r_in = 1.0 if n_in == d else 0.45
                v = rng.standard_normal(n_in).astype(f32)
                v *= f32(per_block * 1.5 / (np.linalg.norm(v) * r_in))             # x 1.5: rows have sigma ~ 1.5 after a few blocks
                W += np.outer(ones, v)
                sd[nm + ".bias"] = sd[nm + ".bias"] + f32(rng.standard_normal() * per_block)
        sd["vision_model.pre_layrnorm.bias"] = sd["vision_model.pre_layrnorm.bias"] + f32(0.5 * common_mode)

    # ---- massive activations
    lm = L // 2
    marker = rng.standard_normal(d).astype(f32)
    marker -= marker.mean()
    marker /= np.linalg.norm(marker)
    patches = rng.choice(np.arange(1, T), size=min(n_marked_patches, T - 1), replace=False)
    marked = np.concatenate([[0], np.sort(patches)]).astype(int)
    out_ch = np.sort(rng.choice(d, size=n_outlier, replace=False))
    pos = sd["vision_model.embeddings.position_embedding.weight"]
    amp = 1.5 * np.sqrt(d)                                      # pre_layrnorm divides it by the row sigma: ~ 0.8 sqrt(d) / ... in the stream
    pos[marked] += f32(amp) * marker
    # the embedding LayerNorm must let the marker through unscaled by its own heavy tail: gain 1 along it is not expressible
    # per channel, so the neuron below is fitted to what actually arrives (marker o gain_pre) instead
    g_pre = sd["vision_model.pre_layrnorm.weight"]
    arrive = marker * g_pre
    arrive_dir = arrive - arrive.mean()
    arrive_dir /= np.linalg.norm(arrive_dir)
    g2, b2 = sd[f"{enc}{lm}.layer_norm2.weight"], sd[f"{enc}{lm}.layer_norm2.bias"]
    fc1w, fc1b = sd[f"{enc}{lm}.mlp.fc1.weight"], sd[f"{enc}{lm}.mlp.fc1.bias"]
    fc2w, fc2b = sd[f"{enc}{lm}.mlp.fc2.weight"], sd[f"{enc}{lm}.mlp.fc2.bias"]
    neurons = rng.choice(mlp, size=2 * n_outlier, replace=False)
    # LayerNorm output along the arrival direction: (x . dir) / sigma_row, ~ N(0, 1) for ordinary tokens; the marked ones sit
    # several sigmas out (measured by tests/diag/adversarial_stats.py).  Threshold 4.5, slope 1.
    w_row = (arrive_dir / g2).astype(f32)                      # (gain o w) = dir
    for k, j in enumerate(neurons):
        sgn = f32(1.0 if k % 2 == 0 else -1.0)
        fc1w[j] = w_row
        fc1b[j] = f32(-4.5 - float(w_row @ b2))
        fc2w[:, j] = 0
        fc2w[out_ch[k // 2], j] = sgn * f32(outlier_gain / 8.0) * f32(1.0 + 0.3 * rng.standard_normal())
    # two neurons per channel with opposite signs would cancel: the second one is given its own sign pattern per channel
    for k in range(n_outlier):
        fc2w[out_ch[k], neurons[2 * k + 1]] = abs(fc2w[out_ch[k], neurons[2 * k + 1]]) * np.sign(fc2w[out_ch[k], neurons[2 * k]])
    fc2b[out_ch] += (rng.choice([-1.0, 1.0], size=n_outlier) * broad_outlier).astype(f32)
    if info is not None:
        info.update(outlier_channels=out_ch.tolist(), marked_tokens=marked.tolist(), outlier_layer=lm, sharp_heads=sharp,
                    marker_dir=arrive_dir)
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
