"""development aid: which geometry makes the fp8 tower produce NaN"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from tests.test_fp8 import _embed
ctx = engine.Context(0)
r = np.random.Generator(np.random.PCG64(0))
base = CLIP_CONFIGS["vit_l14_x2"]
cases = {
    "d1024 T17": dict(base, image_size=56),
    "d1024 T65": dict(base, image_size=112),
    "d1024 T257": base,
    "d768 h12 T257": dict(base, hidden_size=768, num_heads=12, mlp=3072),
    "d512 h8 T257": dict(base, hidden_size=512, num_heads=8, mlp=2048),
    "d1024 mlp2048 T257": dict(base, mlp=2048),
    "d768 T197 p16 x2": dict(CLIP_CONFIGS["vit_b16"], num_layers=2),
    "d1024 h16 T197 p16 x2": dict(CLIP_CONFIGS["vit_b16"], num_layers=2, hidden_size=1024, num_heads=16, mlp=4096),
}
for name, cfg in cases.items():
    sd = random_clip_state_dict(cfg, seed=11, text=False)
    for n in (1, 4):
        pv = r.standard_normal((n, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
        e8 = _embed(engine, ctx, cfg, sd, pv, True)
        e16 = _embed(engine, ctx, cfg, sd, pv, False)
        print(f"{name:24s} n {n}: fp8 NaN {int(np.isnan(e8).sum())} of {e8.size}, bf16 NaN {int(np.isnan(e16).sum())}, 1-cos {float(1 - (e8 * e16).sum(-1).min()):.2e}", flush=True)
