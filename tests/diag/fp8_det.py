"""development aid: which piece of the fp8 tower is not run-to-run deterministic"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from tests.test_fp8 import _gemm, _embed
ctx = engine.Context(0)
r = np.random.Generator(np.random.PCG64(0))
for (M, N, K, kind) in [(1285, 3072, 1024, 0), (1285, 1024, 1024, 0), (1285, 4096, 1024, 1), (1285, 1024, 4096, 0), (1536, 1024, 4096, 0), (5140, 1024, 4096, 0)]:
    A = r.standard_normal((M, K)).astype(np.float32); W = (r.standard_normal((N, K)) * 0.03).astype(np.float32); b = r.standard_normal(N).astype(np.float32)
    outs = [_gemm(ctx, A, W, b, kind)[0] for _ in range(4)]
    print("gemm", M, N, K, kind, "deterministic:", all(np.array_equal(outs[0], o) for o in outs[1:]),
          [int((outs[0] != o).sum()) for o in outs[1:]], flush=True)
cfg = CLIP_CONFIGS["vit_l14_x2"]; sd = random_clip_state_dict(cfg, seed=11, text=False)
for n in (1, 2, 5, 8):
    pv = r.standard_normal((n, 3, 224, 224), dtype=np.float32)
    for rem in (1, 0):
        ctx.set_option("attn_rem", rem)
        e = [_embed(engine, ctx, cfg, sd, pv, True) for _ in range(3)]
        print("vit_l14_x2 n", n, "attn_rem", rem, "deterministic:", all(np.array_equal(e[0], x) for x in e[1:]), [float(np.abs(e[0] - x).max()) for x in e[1:]], flush=True)
ctx.set_option("attn_rem", 1)
cfg = dict(CLIP_CONFIGS["vit_l14_x2"], num_layers=4)
sd = random_clip_state_dict(cfg, seed=11, text=False)
pv = r.standard_normal((5, 3, 224, 224), dtype=np.float32)
e = [_embed(engine, ctx, cfg, sd, pv, True) for _ in range(3)]
print("vit_l14_x4 deterministic:", all(np.array_equal(e[0], x) for x in e[1:]), [float(np.abs(e[0] - x).max()) for x in e[1:]])
