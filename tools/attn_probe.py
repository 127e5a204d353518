"""Two ViT-B/16 layers over 2048 images (one k_attention launch per layer at bench geometry); run under rocprofv3 by tools/attn_ablate.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
ctx = engine.Context(0)
cfg = dict(CLIP_CONFIGS[os.environ.get("ATTN_PROBE_MODEL", "vit_b16")], num_layers=int(os.environ.get("ATTN_PROBE_LAYERS", "2")))
sc = engine.ClipScorer(ctx, cfg, random_clip_state_dict(cfg, seed=6, text=False))
n_img = int(os.environ.get("ATTN_PROBE_IMAGES", "2048"))
pv = np.random.default_rng(0).standard_normal((n_img, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
ctx.set_option("chunk", 4096)
sc.embed_pixels(pv)
sc.embed_pixels(pv)
