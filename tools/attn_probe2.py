"""Round 4: the vision tower's attention kernel at a given geometry and option set, timed with HIP events through the
library (d2r_get_timing covers the whole forward; the attention share comes from rocprofv3 around this script):
python tools/attn_probe2.py <clip> <n_images> [key=value ...]   — three layers, full width."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
name, n = sys.argv[1], int(sys.argv[2])
ctx = engine.Context(0)
cfg = dict(CLIP_CONFIGS[name], num_layers=3)
sc = engine.ClipScorer(ctx, cfg, random_clip_state_dict(cfg, seed=6, text=False))
S = cfg["image_size"]
pv = np.random.default_rng(0).standard_normal((n, 3, S, S), dtype=np.float32)
ctx.set_option("chunk", 4096)
ctx.set_option("cls_last", 0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
ref = sc.embed_pixels(pv)
t0 = time.perf_counter()
for _ in range(3):
    out = sc.embed_pixels(pv)
print(name, n, sys.argv[3:], "wall per forward ms", round((time.perf_counter() - t0) / 3 * 1e3, 2), "checksum", float(np.abs(out).sum()), "identical", bool((out == ref).all()))
np.save(f"/tmp/attn_probe2_{'_'.join(sys.argv[3:]) or 'base'}.npy", out)
