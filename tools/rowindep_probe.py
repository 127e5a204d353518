import sys; sys.path.insert(0, ".")
import numpy as np
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
ctx = engine.Context(0)
cfg = CLIP_CONFIGS["vit_b16"]
sc = engine.ClipScorer(ctx, cfg, random_clip_state_dict(cfg, seed=6, text=False))
pv = np.random.default_rng(0).standard_normal((300, 3, 224, 224), dtype=np.float32)
big = sc.embed_pixels(pv)
for n in (1, 2, 5, 40):
    small = sc.embed_pixels(pv[:n])
    print(n, "bit-identical to the 300-batch rows:", bool((small == big[:n]).all()), float(np.abs(small - big[:n]).max()))
