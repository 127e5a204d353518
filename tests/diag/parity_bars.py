"""Diagnostic (GPU box): measured logit / embedding error of every CLIP test model against the fp32 oracle, on random
frames and on composited frames — the numbers the tolerances in tests/test_gpu_parity.py are set from."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from oracle.pipeline import oracle_logits
from oracle import clip_ref
from tests.parity_utils import cosine, random_unit_text_embeds
from synthetic_scenes import make_scene

ctx = engine.Context(0)
scene = make_scene("shopping")
fg, bg = engine.Testbed(ctx, scene.fg), engine.Testbed(ctx, scene.bg)
from dream2real_amd.accio2ngp import converter
for name, n in (("vit_tiny", 24), ("vit_b16", 8), ("vit_l14_x2", 6), ("vit_l14_336_x1", 4), ("vit_l14", 4), ("vit_l14_336", 4)):
    cfg = CLIP_CONFIGS[name]
    sd = random_clip_state_dict(cfg, seed=6, text=False)
    sc = engine.ClipScorer(ctx, cfg, sd)
    r = np.random.Generator(np.random.PCG64(3))
    text = random_unit_text_embeds(cfg["proj"], 3)
    frames = r.integers(0, 256, size=(n, 90, 160, 3), dtype=np.uint8)
    t = time.time()
    lg, emb = sc.score_frames(frames, text, return_embeds=True)
    olg, oemb = oracle_logits(frames, cfg, sd, text)
    de = emb - oemb
    print(f"{name:16s} random frames n={n}: max|dlogit|/scale {np.abs(lg - olg).max() / sc.logit_scale:.2e}  1-cos max {(1 - cosine(emb, oemb)).max():.2e}  "
          f"|de| max {np.linalg.norm(de, axis=1).max():.2e}  |de|/sqrt(D) {np.linalg.norm(de, axis=1).max() / np.sqrt(cfg['proj']):.2e}  ({time.time() - t:.1f}s)", flush=True)
    sc.close()
