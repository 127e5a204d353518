"""the 1e-3 parity bar under a TRAINED checkpoint's statistics.  Every ViT parity number of rounds 1-4
was measured on Gaussian weights (clip_model.random_clip_state_dict); `adversarial_clip_state_dict` adds massive activation
channels, heavy-tailed LayerNorm gains, a per-token common mode of the residual stream and near one-hot attention heads
(tests/diag/adversarial_stats.py prints the realised statistics).  This module measures |dlogit| / logit_scale of the FUSED
product path (d2r_render_score_host: layer-0 reuse and the class-token-only last block on, as shipped) against the fp32 oracle
on composited frames, for every vision-tower schedule (`ln_fold` 0..4) — `measure()` is what tests/test_gpu_parity.py asserts
on; run as a script it sweeps the three encoders on >= 64 frames each and writes profiles/r05_adversarial_parity.{json,md}.

Test infrastructure: imports oracle/ as the checker."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from dream2real_amd.clip_model import CLIP_CONFIGS, adversarial_clip_state_dict, random_clip_state_dict  # noqa: E402
from oracle import host_ref  # noqa: E402
from oracle.pipeline import oracle_logits  # noqa: E402

MODES = (0, 1, 2, 3, 4)


def distinct_candidate_poses(scene, n, grid=None):
    """n candidate object poses (NGP convention, [n,4,4]) spread over the scene type's grid bounds"""
    g = grid or int(np.ceil(np.sqrt(n * 1.5)))
    poses = host_ref.sample_poses_grid(scene.scene_centre, [g, g, 1, 1, 1, 1], scene.scene_type).reshape(-1, 4, 4)
    idx = np.linspace(0, len(poses) - 1, n).astype(int)
    return host_ref.converter(poses[idx])


def measure(engine, ctx, scene, fg, bg, name, W, H, n_frames, *, weights="adversarial", modes=MODES, seed=6, n_text=3,
            variants=(), oracle_batch=16, gen_kwargs=None, floor_frames=0, log=print):
    """-> dict: per ln_fold mode (and per named variant = dict of context options, run at the default mode) the max and rms of
    |dlogit| / logit_scale and the max of 1 - cos(embedding) over n_frames composited frames x n_text captions."""
    from tests.parity_utils import cosine, random_unit_text_embeds
    cfg = CLIP_CONFIGS[name]
    sd = adversarial_clip_state_dict(cfg, seed, **(gen_kwargs or {})) if weights == "adversarial" else random_clip_state_dict(cfg, seed, text=False)
    sc = engine.ClipScorer(ctx, cfg, sd)
    cam = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    bg_rgba, bg_depth = bg.render_batch(cam[None, :3], W, H)
    view = fg.view(W, H)
    ctx.set_background(view, bg_rgba[0], bg_depth[0])
    poses = distinct_candidate_poses(scene, n_frames)
    text = random_unit_text_embeds(cfg["proj"], n_text)
    out = {"model": name, "weights": weights, "frames": n_frames, "size": [W, H], "captions": n_text, "modes": {}, "variants": {}}
    frames0 = None
    runs = [("mode", m, {"ln_fold": m}) for m in modes] + [("variant", k, dict(v)) for k, v in variants]
    defaults = {"ln_fold": 4, "l0_reuse": 1, "cls_last": 1, "prep_reuse": 1}
    results = []
    try:
        for kind, key, opts in runs:
            for k, v in {**defaults, **opts}.items():
                ctx.set_option(k, v)
            logits, frames = engine.render_score_host(ctx, fg, sc, view, T1, cam, poses, text, return_frames=True)
            if frames0 is None:
                frames0 = frames
            else:
                np.testing.assert_array_equal(frames, frames0)            # the render does not depend on the tower's schedule
            results.append((kind, key, logits))
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)
    t0 = time.time()
    olg_parts, oemb_parts = [], []
    for j in range(0, n_frames, oracle_batch):                          # bounded host memory: attention scores are B x H x T x T floats
        a, b = oracle_logits(frames0[j:j + oracle_batch], cfg, sd, text)
        olg_parts.append(a)
        oemb_parts.append(b)
    olg, oemb = np.concatenate(olg_parts), np.concatenate(oemb_parts)
    out["oracle_seconds"] = round(time.time() - t0, 1)
    if floor_frames:
        # the ideal-bf16 tower (oracle/clip_bf16.py) on the first frames: how far ANY bf16-operand implementation is from fp32 here
        from oracle import clip_bf16, render_ref
        k = min(floor_frames, n_frames)
        pv = np.stack([render_ref.clip_preprocess(f, cfg["image_size"], True)[0] for f in frames0[:k]])
        d = np.abs((clip_bf16.vision_embeds(pv, sd, cfg) - oemb[:k]) @ text.T)
        out["ideal_bf16_floor"] = {"frames": k, "max": float(d.max()), "rms": float(np.sqrt((d ** 2).mean()))}
        log(f"[adversarial parity] {name} {weights}: ideal-bf16 floor on the first {k} frames: max {d.max():.2e} rms {np.sqrt((d ** 2).mean()):.2e}")
    out["distinct_frames"] = len({f.tobytes() for f in frames0})
    scale = float(sc.logit_scale)
    for kind, key, logits in results:
        d = np.abs(logits - olg) / scale
        rec = {"max": float(d.max()), "rms": float(np.sqrt((d ** 2).mean())), "p99": float(np.quantile(d, 0.99))}
        out["modes" if kind == "mode" else "variants"][str(key)] = rec
        log(f"[adversarial parity] {name} {weights} {W}x{H} n={n_frames}: {kind} {key}: |dlogit|/scale max {rec['max']:.2e} rms {rec['rms']:.2e}  (bar 1e-3)")
    # the embeddings of the default mode, for 1 - cos
    sc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="vit_b16,vit_l14,vit_l14_336")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--floor-frames", type=int, default=0, help="also run the ideal-bf16 tower (oracle/clip_bf16.py) on this many frames")
    ap.add_argument("--benign", action="store_true", help="also run the Gaussian weights through the same measurement")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r05_adversarial_parity.json"))
    a = ap.parse_args()
    from dream2real_amd import engine
    from tests.scenes import make_scene
    scene = make_scene("shopping")
    ctx = engine.Context(0)
    fg = engine.Testbed(ctx, scene.fg)
    fg.background_color = list(scene.fg_background)
    bg = engine.Testbed(ctx, scene.bg)
    sizes = {"vit_b16": (640, 360), "vit_l14": (640, 360), "vit_l14_336": (336, 336)}
    variants = (("l0_reuse=0", {"l0_reuse": 0}), ("cls_last=0", {"cls_last": 0}), ("l0_reuse=0,cls_last=0", {"l0_reuse": 0, "cls_last": 0}))
    res = []
    for name in a.models.split(","):
        W, H = sizes.get(name, (640, 360))
        for w in (("adversarial", "benign") if a.benign else ("adversarial",)):
            res.append(measure(engine, ctx, scene, fg, bg, name, W, H, a.n, weights=w, variants=variants, floor_frames=a.floor_frames))
            os.makedirs(os.path.dirname(a.out), exist_ok=True)
            json.dump(res, open(a.out, "w"), indent=1)
    md = ["| model | weights | frames | " + " | ".join(f"ln_fold {m}" for m in MODES) + " | " + " | ".join(k for k, _ in variants) + " |",
          "|---|---|---|" + "---|" * (len(MODES) + len(variants))]
    for r in res:
        md.append(f"| {r['model']} {r['size'][0]}x{r['size'][1]} | {r['weights']} | {r['frames']} ({r['distinct_frames']} distinct) | " +
                  " | ".join(f"{r['modes'][str(m)]['max']:.2e} ({r['modes'][str(m)]['rms']:.1e})" for m in MODES) + " | " +
                  " | ".join(f"{r['variants'][k]['max']:.2e}" for k, _ in variants) + " |")
    open(a.out.replace(".json", ".md"), "w").write(
        "max (rms) of |dlogit| / logit_scale against the fp32 oracle, fused path (d2r_render_score_host), bar 1e-3\n\n" + "\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
