#!/usr/bin/env python3
"""Stage-by-stage HIP-vs-oracle error report (run on the GPU box).  Prints, does not assert;
the thresholds in tests/test_gpu_parity.py come from these numbers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from dream2real_amd import engine
from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from tests.scenes import make_scene
from oracle import clip_ref, host_ref, render_ref
from tests.parity_utils import OraclePipeline, cosine, oracle_logits, random_unit_text_embeds


def main():
    which = sys.argv[1:] or ["field", "render", "composite", "prep", "vit_tiny", "vit_b16"]
    scene = make_scene("shopping")
    ctx = engine.Context(0)
    fg = engine.Testbed(ctx, scene.fg)
    bg = engine.Testbed(ctx, scene.bg)
    ofg = render_ref.OracleNerf(scene.fg)
    r = np.random.Generator(np.random.PCG64(0))

    if "field" in which:
        n = 4096
        occ = np.argwhere(scene.fg.occupancy_bool())            # z,y,x
        cells = occ[r.integers(0, len(occ), n)]
        xyz = ((cells[:, ::-1] + r.random((n, 3))) / 128.0).astype(np.float32)
        d = r.standard_normal((n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        got = fg.eval_points(xyz, d)
        want = render_ref.eval_points(ofg, xyz, d)
        rel_s = np.abs(got[:, 0] - want[:, 0]) / want[:, 0]
        print(f"[field] sigma rel err: max {rel_s.max():.3e} mean {rel_s.mean():.3e};  rgb abs err: max "
              f"{np.abs(got[:, 1:] - want[:, 1:]).max():.3e} mean {np.abs(got[:, 1:] - want[:, 1:]).mean():.3e}")
        print("        sample", got[:2], want[:2])

    W, H = 160, 90
    pipe = OraclePipeline(scene, W, H)
    poses = host_ref.sample_poses_grid(scene.scene_centre, [3, 2, 1, 1, 1, 1], scene.scene_type).reshape(-1, 4, 4)
    if "render" in which:
        cams = np.stack([pipe.fg_camera(p) for p in poses])
        fg.background_color = list(scene.fg_background)
        t = time.time()
        rgba, depth = fg.render_batch(cams, W, H)
        print(f"[render] HIP {len(cams)} frames {time.time() - t:.3f}s samples {fg.last_samples}")
        ns0 = pipe.n_samples
        for i, p in enumerate(poses):
            orgba, odepth = pipe.fg_render(p)
            hit_g, hit_o = depth[i] > 0, odepth > 0
            print(f"   cam {i}: hit px HIP {hit_g.sum()} oracle {hit_o.sum()} mismatch {(hit_g != hit_o).sum()}; "
                  f"max|drgba| {np.abs(rgba[i] - orgba).max():.3e} mean {np.abs(rgba[i] - orgba)[hit_o].mean():.3e}; "
                  f"max|ddepth| {np.abs(depth[i] - odepth).max():.3e}")
        print(f"   oracle samples {pipe.n_samples - ns0}")
        # background
        cam_bg = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
        t = time.time()
        brgba, bdepth = bg.render_batch(cam_bg[None, :3], W, H)
        print(f"[render] bg HIP {time.time() - t:.3f}s samples {bg.last_samples}")
        orgba, odepth = pipe.background()
        print(f"   bg: max|drgba| {np.abs(brgba[0] - orgba).max():.3e} mean {np.abs(brgba[0] - orgba).mean():.3e} "
              f"max|ddepth| {np.abs(bdepth[0] - odepth).max():.3e} hit mismatch {((bdepth[0] > 0) != (odepth > 0)).sum()}")

    if "composite" in which:
        fg.background_color = list(scene.fg_background)
        view = fg.view(W, H)
        obg = pipe.background()
        ctx.set_background(view, obg[0], obg[1])     # same background on both sides
        T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
        TC = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
        frames = fg.render_composite(view, T1, TC, host_ref.converter(poses.astype(np.float32)))
        want = pipe.frames(poses, bg=obg)
        diff = np.abs(frames.astype(int) - want.astype(int))
        print(f"[composite] max LSB diff {diff.max()}, px>1LSB {(diff.max(-1) > 1).mean() * 100:.4f}%, "
              f"px>0 {(diff.max(-1) > 0).mean() * 100:.4f}%  stats {ctx.render_stats()}")
        np.save("gpurun_out/frames_hip.npy", frames)
        np.save("gpurun_out/frames_oracle.npy", want)

    if "prep" in which:
        cfg = CLIP_CONFIGS["vit_b16"]
        sd = random_clip_state_dict(dict(cfg, num_layers=1), seed=6, text=False)
        sc = engine.ClipScorer(ctx, dict(cfg, num_layers=1), sd)
        for (hh, ww) in ((360, 640), (90, 160), (336, 336), (224, 224)):
            f = r.integers(0, 256, size=(2, hh, ww, 3), dtype=np.uint8)
            got = sc.preprocess(f, rot90=True)
            want = np.stack([render_ref.clip_preprocess(x, cfg["image_size"], True)[0] for x in f])
            print(f"[prep] {ww}x{hh}: max|dpv| {np.abs(got - want).max():.3e}")
        sc.close()

    for name in ("vit_tiny", "vit_b16"):
        if name not in which:
            continue
        cfg = CLIP_CONFIGS[name]
        sd = random_clip_state_dict(cfg, seed=6, text=False)
        sc = engine.ClipScorer(ctx, cfg, sd)
        n = 6
        pv = r.standard_normal((n, 3, cfg["image_size"], cfg["image_size"]), dtype=np.float32)
        t = time.time()
        got = sc.embed_pixels(pv)
        t1 = time.time() - t
        want = clip_ref.vision_embeds(pv, sd, cfg)
        print(f"[{name}] cos(HIP, oracle) min {cosine(got, want).min():.6f}; max|d| {np.abs(got - want).max():.3e}; "
              f"cross-image cos(oracle) {cosine(want[0], want[1]):.4f}  ({t1:.3f}s)")
        text = random_unit_text_embeds(cfg["proj"])
        frames = r.integers(0, 256, size=(n, 90, 160, 3), dtype=np.uint8)
        lg = sc.score_frames(frames, text)
        olg, _ = oracle_logits(frames, cfg, sd, text)
        print(f"   logits max|d| {np.abs(lg - olg).max():.3e} (scale {sc.logit_scale:.1f}) -> cosine err {np.abs(lg - olg).max() / sc.logit_scale:.3e}")
        sc.close()


if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    main()
