"""Generates tests/golden/hf_clip_r05.npz — run in the BUILD container only (imports Hugging Face transformers; nothing
here travels as code, only the data file does).  Round 5 additions to the g5 family of make_goldens.py (reference
clip_scoring.py:150,180-181: `CLIPModel.forward`):

  g5_vit_l14_*        full-depth ViT-L/14 (224 px, 24 layers, d 1024, 257 tokens — BASELINE.json configs[4]'s encoder) with
                      the seeded Gaussian weights: image embeddings of 2 seeded inputs (g5 held the 2-layer model and
                      ViT-B/16 only)
  g5adv_<model>_*     the SAME Hugging Face forward under `adversarial_clip_state_dict` (massive activation channels,
                      heavy-tailed and partly negative LayerNorm gains, common mode, near one-hot heads) for vit_tiny and
                      vit_b16: pins the fp32 oracle (oracle/clip_ref.py) in the regime the round-5 parity tests use it in

Inputs are regenerated from their seeds by the tests (PCG64(77) standard normal); the file holds the embeddings
(float32) and a SHA-256 of their bytes."""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from transformers import CLIPConfig, CLIPModel  # noqa: E402

from dream2real_amd.clip_model import CLIP_CONFIGS, adversarial_clip_state_dict, random_clip_state_dict  # noqa: E402


def hf_vision(cfg, sd):
    c = CLIPConfig(
        vision_config=dict(hidden_size=cfg["hidden_size"], intermediate_size=cfg["mlp"], num_hidden_layers=cfg["num_layers"],
                           num_attention_heads=cfg["num_heads"], image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                           projection_dim=cfg["proj"], hidden_act="quick_gelu"),
        text_config=dict(hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=1, vocab_size=64,
                         max_position_embeddings=8, projection_dim=cfg["proj"], hidden_act="quick_gelu"),
        projection_dim=cfg["proj"])
    m = CLIPModel(c).eval()
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith(("vision_model.", "visual_projection."))}
    missing, unexpected = m.load_state_dict(t, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("text_model.", "text_projection.", "logit_scale")) or "position_ids" in k for k in missing), missing
    return m


def embeds(m, pv):
    with torch.no_grad():
        e = m.get_image_features(pixel_values=torch.from_numpy(pv))
        if not torch.is_tensor(e):                    # transformers 5.x returns an output object
            e = e.pooler_output
    e = e / e.norm(dim=-1, keepdim=True)
    return e.numpy().astype(np.float32)


def main():
    torch.manual_seed(0)
    out = {}
    for key, name, weights, n in (("g5_vit_l14", "vit_l14", "gaussian", 2), ("g5adv_vit_tiny", "vit_tiny", "adversarial", 4),
                                  ("g5adv_vit_b16", "vit_b16", "adversarial", 2)):
        cfg = CLIP_CONFIGS[name]
        sd = random_clip_state_dict(cfg, 6, text=False) if weights == "gaussian" else adversarial_clip_state_dict(cfg, 6)
        r = np.random.Generator(np.random.PCG64(77))
        pv = r.standard_normal((n, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
        e = embeds(hf_vision(cfg, sd), pv)
        out[key + "_image_embeds"] = e
        out[key + "_sha256"] = np.frombuffer(hashlib.sha256(e.tobytes()).digest(), np.uint8)
        print(key, e.shape, hashlib.sha256(e.tobytes()).hexdigest()[:16], e[0, :4])
    np.savez_compressed(os.path.join(HERE, "hf_clip_r05.npz"), **out)
    print("wrote hf_clip_r05.npz", os.path.getsize(os.path.join(HERE, "hf_clip_r05.npz")), "bytes")


if __name__ == "__main__":
    main()
