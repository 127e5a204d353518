"""VERDICT r04 next #1, NeRF side: the marcher against d2r_oracle_render on a field with TRAINED-like statistics
(synthetic_scenes.make_trained_like_nerf: table values to +-8, density pre-activations over +-12, an opaque shell about three
march steps thick) instead of the Xavier / U(-0.5, 0.5) one every earlier parity number used.  measure() returns
  field   : |d log sigma| and |d rgb| of Testbed.eval_points against the oracle on points around the shell
  frames  : composited uint8 frames, HIP vs oracle (same background): share of pixels off by 0 / 1 / more LSB, max
  logits  : END TO END — oracle render + oracle CLIP (fp32) against the fused HIP path, |dlogit| / logit_scale
Run as a script it prints them for the trained-like and the plain shopping scene side by side.
Test infrastructure: imports oracle/ as the checker."""
from __future__ import annotations

import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from dream2real_amd.clip_model import CLIP_CONFIGS, adversarial_clip_state_dict, random_clip_state_dict  # noqa: E402
from dream2real_amd.scene import world_to_ngp  # noqa: E402
from oracle import host_ref, render_ref  # noqa: E402
from oracle.pipeline import OraclePipeline, oracle_logits  # noqa: E402


def measure(engine, ctx, kind="shopping_trained", W=160, H=90, grid=(4, 3, 2), clip="vit_b16", clip_weights="benign", mlp_f16=0, log=print):
    from tests.scenes import make_scene
    from tests.parity_utils import random_unit_text_embeds
    was = ctx.get_option("mlp_f16")
    ctx.set_option("mlp_f16", mlp_f16)
    try:
        return _measure(engine, ctx, kind, W, H, grid, clip, clip_weights, mlp_f16, log)
    finally:
        ctx.set_option("mlp_f16", was)


def _measure(engine, ctx, kind, W, H, grid, clip, clip_weights, mlp_f16, log):
    from tests.scenes import make_scene
    from tests.parity_utils import random_unit_text_embeds
    scene = make_scene(kind)
    fg = engine.Testbed(ctx, scene.fg)
    fg.background_color = list(scene.fg_background)
    out = {"scene": kind, "size": [W, H], "mlp": "fp16" if mlp_f16 else "bf16"}
    kind = f"{kind} [{out['mlp']} MLP]"
    # ---- field
    r = np.random.Generator(np.random.PCG64(0))
    n = 20000
    c = world_to_ngp(scene.obj_pose[:3, 3])
    occ = np.argwhere(scene.fg.occupancy_bool())
    cells = occ[r.integers(0, len(occ), n)]
    xyz = ((cells[:, ::-1] + r.random((n, 3))) / 128.0).astype(np.float32)
    d = r.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = fg.eval_points(xyz, d)
    want = render_ref.eval_points(render_ref.OracleNerf(scene.fg), xyz, d)
    assert np.isfinite(got).all()
    dls = np.abs(np.log(np.maximum(got[:, 0], 1e-30)) - np.log(np.maximum(want[:, 0], 1e-30)))
    act = (want[:, 0] * 0.0016914558 > 1e-4) & (want[:, 0] * 0.0016914558 < 30.0)            # where alpha is neither 0 nor saturated
    out["field"] = {"log_sigma_range": [float(np.log(want[:, 0]).min()), float(np.log(want[:, 0]).max())],
                    "dlog_sigma_max_active": float(dls[act].max()), "dlog_sigma_p999_active": float(np.quantile(dls[act], 0.999)),
                    "dlog_sigma_rms_active": float(np.sqrt((dls[act] ** 2).mean())), "active_share": float(act.mean()),
                    "drgb_max": float(np.abs(got[:, 1:] - want[:, 1:]).max())}
    log(f"[trained field] {kind}: log sigma in [{out['field']['log_sigma_range'][0]:.1f}, {out['field']['log_sigma_range'][1]:.1f}]; where alpha is live "
        f"({act.mean():.0%} of points) |dlog sigma| max {out['field']['dlog_sigma_max_active']:.3f} p99.9 {out['field']['dlog_sigma_p999_active']:.3f} "
        f"rms {out['field']['dlog_sigma_rms_active']:.4f}; |drgb| max {out['field']['drgb_max']:.4f}")
    # ---- frames
    pipe = OraclePipeline(scene, W, H)
    poses = host_ref.sample_poses_grid(scene.scene_centre, list(grid) + [1, 1, 1], scene.scene_type).reshape(-1, 4, 4)
    obg = pipe.background()
    view = fg.view(W, H)
    ctx.set_background(view, obg[0], obg[1])
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    TC = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    want_frames = pipe.frames(poses, bg=obg)
    cfg = CLIP_CONFIGS[clip]
    sd = adversarial_clip_state_dict(cfg, 6) if clip_weights == "adversarial" else random_clip_state_dict(cfg, 6, text=False)
    sc = engine.ClipScorer(ctx, cfg, sd)
    text = random_unit_text_embeds(cfg["proj"], 3)
    logits, frames = engine.render_score_host(ctx, fg, sc, view, T1, TC, host_ref.converter(poses.astype(np.float32)), text, return_frames=True)
    diff = np.abs(frames.astype(int) - want_frames.astype(int)).max(-1)
    changed = (want_frames != want_frames[:1]).any(-1) | (frames != want_frames[:1]).any(-1)
    hit = (want_frames != np.broadcast_to(_bg_u8(pipe, obg), want_frames.shape)).any(-1)
    out["frames"] = {"n": len(poses), "max": int(diff.max()), "share_off_by_1": float((diff == 1).mean()), "share_off_by_more": float((diff > 1).mean()),
                     "object_pixels": int(hit.sum()), "object_share_off_by_more": float((diff[hit] > 1).mean()) if hit.any() else 0.0,
                     "samples_oracle": int(pipe.n_samples)}
    del changed
    log(f"[trained field] {kind} {W}x{H}, {len(poses)} frames: max |d| {out['frames']['max']} LSB, off by 1: {out['frames']['share_off_by_1']:.4%}, "
        f"by more: {out['frames']['share_off_by_more']:.4%} of all pixels = {out['frames']['object_share_off_by_more']:.2%} of the {hit.sum()} object pixels")
    # ---- end to end: oracle frames -> oracle CLIP  vs  HIP frames -> HIP CLIP
    olg, _ = oracle_logits(want_frames, cfg, sd, text)
    d_e2e = np.abs(logits - olg) / float(sc.logit_scale)
    olg2, _ = oracle_logits(frames, cfg, sd, text)                       # the tower alone, on the HIP frames
    d_vit = np.abs(logits - olg2) / float(sc.logit_scale)
    d_render = np.abs(olg2 - olg) / float(sc.logit_scale)                 # what the frame differences alone do to a logit (fp32 tower both sides)
    out["logits"] = {"clip": clip, "clip_weights": clip_weights, "end_to_end_max": float(d_e2e.max()), "tower_only_max": float(d_vit.max()),
                     "render_only_max": float(d_render.max())}
    log(f"[trained field] {kind} + {clip} ({clip_weights} weights): |dlogit|/scale end to end {d_e2e.max():.2e} = tower {d_vit.max():.2e} (+) render {d_render.max():.2e}  (bar 1e-3)")
    sc.close()
    fg.close()
    return out


def measure_distances(engine, ctx, kind="shopping_trained", W=160, H=90, grid=(6, 4, 1), clip="vit_b16", log=print):
    """how far each arithmetic sits from an EMULATION of tiny-cuda-nn's half arithmetic (oracle arith mode 1: half
    corner accumulation in the grid, half accumulators and half inter-layer activations in both MLPs; mode 2 = the grid's newer
    fma form) on the trained-like field — the fp32 specification (oracle mode 0), the HIP marcher with bf16 MLP operands (north_star)
    and with fp16 operands (option mlp_f16), all with fp32 accumulation.  Per contender: field (|dlog sigma|, |drgb|), composited
    frames (pixels off by 1 / more LSB, share of the object's pixels) and logits through the SAME fp32 tower (so only the render
    differs).  Returns {"emulation": .., "rows": {name: {...}}}."""
    from tests.scenes import make_scene
    from tests.parity_utils import random_unit_text_embeds
    scene = make_scene(kind)
    fg = engine.Testbed(ctx, scene.fg)
    fg.background_color = list(scene.fg_background)
    r = np.random.Generator(np.random.PCG64(0))
    n = 20000
    occ = np.argwhere(scene.fg.occupancy_bool())
    cells = occ[r.integers(0, len(occ), n)]
    xyz = ((cells[:, ::-1] + r.random((n, 3))) / 128.0).astype(np.float32)
    d = r.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    om = render_ref.OracleNerf(scene.fg)
    pipe = OraclePipeline(scene, W, H)
    poses = host_ref.sample_poses_grid(scene.scene_centre, list(grid) + [1, 1, 1], scene.scene_type).reshape(-1, 4, 4)
    obg = pipe.background()                                  # one background for every contender (an input of the composite)
    view = fg.view(W, H)
    ctx.set_background(view, obg[0], obg[1])
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    TC = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    pn = host_ref.converter(poses.astype(np.float32))
    cfg = CLIP_CONFIGS[clip]
    sd = random_clip_state_dict(cfg, 6, text=False)
    text = random_unit_text_embeds(cfg["proj"], 3)
    field, frames = {}, {}
    for name, mode in (("emulation", 1), ("emulation_fma_grid", 2), ("fp32 specification (oracle)", 0)):
        old = render_ref.set_arith(mode)
        try:
            field[name] = render_ref.eval_points(om, xyz, d)
            frames[name] = pipe.frames(poses, bg=obg)
        finally:
            render_ref.set_arith(old)
    was = ctx.get_option("mlp_f16")
    for name, f16 in (("HIP, bf16 MLP operands", 0), ("HIP, fp16 MLP operands (mlp_f16)", 1)):
        ctx.set_option("mlp_f16", f16)
        try:
            field[name] = fg.eval_points(xyz, d)
            frames[name] = fg.render_composite(view, T1, TC, pn)
        finally:
            ctx.set_option("mlp_f16", was)
    ref_f, ref_fr = field["emulation"], frames["emulation"]
    hit = (ref_fr != np.broadcast_to(_bg_u8(pipe, obg), ref_fr.shape)).any(-1)
    act = (ref_f[:, 0] * 0.0016914558 > 1e-4) & (ref_f[:, 0] * 0.0016914558 < 30.0)
    scale = float(np.exp(np.float32(sd.get("logit_scale", 4.6052))))
    lg_ref, _ = oracle_logits(ref_fr, cfg, sd, text)
    rows = {}
    for name in field:
        if name == "emulation":
            continue
        dls = np.abs(np.log(np.maximum(field[name][:, 0], 1e-30)) - np.log(np.maximum(ref_f[:, 0], 1e-30)))
        diff = np.abs(frames[name].astype(int) - ref_fr.astype(int)).max(-1)
        lg, _ = oracle_logits(frames[name], cfg, sd, text)
        rows[name] = {"dlog_sigma_max": float(dls[act].max()), "dlog_sigma_rms": float(np.sqrt((dls[act] ** 2).mean())),
                      "drgb_max": float(np.abs(field[name][:, 1:] - ref_f[:, 1:]).max()),
                      "pixels_off_by_1": float((diff == 1).mean()), "pixels_off_by_more": float((diff > 1).mean()),
                      "object_pixels_off_by_more": float((diff[hit] > 1).mean()) if hit.any() else 0.0,
                      "logit_max": float(np.abs(lg - lg_ref).max() / scale)}
        log(f"[fp16-accumulation emulation] {name:36s}: |dlog sigma| max {rows[name]['dlog_sigma_max']:.4f} rms {rows[name]['dlog_sigma_rms']:.4f}  |drgb| {rows[name]['drgb_max']:.4f}  "
            f"pixels off by 1 {rows[name]['pixels_off_by_1']:.3%} by more {rows[name]['pixels_off_by_more']:.3%} ({rows[name]['object_pixels_off_by_more']:.2%} of the object's)  "
            f"|dlogit|/scale {rows[name]['logit_max']:.2e}")
    fg.close()
    return {"scene": kind, "size": [W, H], "frames": int(len(poses)), "object_pixels": int(hit.sum()), "clip": clip, "rows": rows}


def _bg_u8(pipe, obg):
    """the oracle's composite of an EMPTY foreground over the background = the background frame in uint8"""
    z = np.zeros_like(obg[0])
    return render_ref.composite(z, np.zeros_like(obg[1]), obg[0], obg[1])[None]


if __name__ == "__main__":
    import json
    from dream2real_amd import engine
    ctx = engine.Context(0)
    res = []
    for f16 in (0, 1):
        for kind in ("shopping_trained", "shopping"):
            res.append(measure(engine, ctx, kind, 160, 90, (6, 4, 2), "vit_b16", "benign", mlp_f16=f16))
        res.append(measure(engine, ctx, "shopping_trained", 640, 360, (3, 2, 1), "vit_b16", "benign", mlp_f16=f16))
    res.append(measure(engine, ctx, "shopping_trained", 160, 90, (6, 4, 2), "vit_b16", "adversarial"))
    res.append({"distances_to_fp16_accumulation_emulation": [measure_distances(engine, ctx, "shopping_trained", 160, 90, (6, 4, 1)),
                                                              measure_distances(engine, ctx, "shopping", 160, 90, (6, 4, 1))]})
    out = os.path.join(REPO, "gpurun_out", "r06_trained_field_parity.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    ctx.close()
