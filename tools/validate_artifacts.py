#!/usr/bin/env python3
"""validate_artifacts.py — parity against the REAL artefacts, the day they are at hand.

Nothing the reference's results depend on is available offline: no instant-ngp snapshot, no OpenAI
CLIP weights / vocabulary, no output of a reference run.  Everywhere else this repo stands on restatements pinned by
goldens; this tool is the harness that turns "parity unpinned" into a measured number as soon as a directory of the
reference's own files exists (reference install.sh:38-50 downloads `method_out/<scene>/`; dream2real.py:356-358 and
combined_rendering.py:157-159 write it):

  method_out/<scene>/fg_base.ingp, bg_base.ingp     snapshots (reconstruction/ngp_visual_model.py:24-28)
  method_out/<scene>/cb_render/cb_rgb_%04d.png      composited renders of the valid poses, in pose order
  method_out/<scene>/pose_batch.txt, pose_scores.txt, goal_pose.txt      np.savetxt of optimise_pose_grid's results

and a Hugging Face CLIP checkpoint directory (model.safetensors, vocab.json, merges.txt; clip_scoring.py:150-151).

Sections (each runs when its inputs are present and says what it skipped otherwise):
  (a) snapshots   every key of the msgpack tree the reader ignores, the parameter / density-grid counts and the level
                  table the reader derives next to what the file holds (d2r_ingp_inspect: host only, runs without a
                  GPU), then the GPU loader's verdict on the file
  (b) clip        re-score cb_render/*.png with the supplied weights through d2r_clip_score_frames and compare with
                  pose_scores.txt — CLIP-only parity, no NeRF involved.  The file holds smoothed scores when the run
                  smoothed (the demos do): both the raw ratio and the smoothed scores are compared, the better match
                  is reported with its error
  (c) render      render pose_batch.txt's valid poses with the snapshots and compare with the PNGs: PSNR and histogram
                  of |difference| in LSB per frame (needs the movable object's pose and the view's camera pose, which
                  the reference recomputes from the scene instead of caching: --obj-pose / --cam-pose, 4x4 np.savetxt
                  files in the reference's world / OpenCV convention)
  (d) argmax      the best pose from (b)'s scores against goal_pose.txt

Prints a JSON report (also to --out).  Exit code 0 = every section that ran is within its bar, 1 = something is off,
2 = nothing could run.  The product path only: nothing here touches oracle/.
"""
import argparse
import glob
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def inspect_snapshot(path: str) -> dict:
    """Section (a), host side: what the file holds vs what the loader reads and derives."""
    from dream2real_amd import _lib
    data = open(path, "rb").read()
    text = _lib.ingp_inspect(data)
    ignored, read, checked, unknown, derived = [], [], [], [], {}
    for ln in text.splitlines():
        if ln.startswith("# derived:"):
            t = ln.split()[2:]
            derived.update({t[i]: int(t[i + 1]) for i in range(0, len(t) - 1, 2) if t[i + 1].lstrip("-").isdigit()})
        elif ln.startswith("# level "):
            derived.setdefault("levels", []).append(ln[2:])
        elif ln.startswith("- "):
            ignored.append(ln[2:])
        elif ln.startswith("R "):
            read.append(ln[2:])
        elif ln.startswith("C "):
            checked.append(ln[2:])
        elif ln.startswith("? "):
            unknown.append(ln[2:])
    problems = []
    try:                                   # the loader's own verdict (every check d2r_nerf_load_ingp makes, host only)
        _lib.ingp_validate(data)
    except _lib.D2RError as e:
        problems.append(f"d2r_nerf_load_ingp would refuse the file: {e}")
    if "n_params_expected" in derived and derived["n_params_expected"] != derived.get("params_binary_halves"):
        problems.append(f"params_binary holds {derived.get('params_binary_halves')} halves, the reader derives {derived['n_params_expected']}")
    if "density_grid_halves_expected" in derived and derived["density_grid_halves_expected"] != derived.get("density_grid_binary_halves"):
        problems.append(f"density_grid_binary holds {derived.get('density_grid_binary_halves')} halves, the reader expects "
                        f"{derived['density_grid_halves_expected']}")
    if not derived:
        problems.append("the encoding fields are outside what the reader accepts")
    return {"file": path, "keys_read": read, "keys_checked": checked, "keys_ignored": ignored, "keys_unknown": unknown, "derived": derived,
            "problems": problems}


def load_frames(render_dir: str):
    from PIL import Image
    files = sorted(f for f in os.listdir(render_dir) if f.lower().endswith(".png"))      # the reference sorts os.listdir (clip_scoring.py:98)
    return [np.asarray(Image.open(os.path.join(render_dir, f)).convert("RGB")) for f in files], files


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1]) if len(a) > 2 else 1.0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--method-out", required=True, help="method_out/<scene>/ of a reference run")
    ap.add_argument("--clip", default=None, help="Hugging Face CLIP checkpoint directory (model.safetensors, vocab.json, merges.txt)")
    ap.add_argument("--goal-caption", default=None)
    ap.add_argument("--norm-caption", action="append", default=None, help="normalising caption (repeatable)")
    ap.add_argument("--captions-json", default=None, help="json with {scene: {goal_caption, norm_captions}} (tests/golden/captions.json)")
    ap.add_argument("--scene", default=None, help="key into --captions-json")
    ap.add_argument("--sample-res", default=None, help="x,y,z,rx,ry,rz of the pose grid (for the smoothed comparison)")
    ap.add_argument("--obj-pose", default=None, help="4x4 txt: the movable object's pose T_WO_1 (world)")
    ap.add_argument("--cam-pose", default=None, help="4x4 txt: the render view's camera pose (OpenCV convention, as opt_cam_poses)")
    ap.add_argument("--view-idx", type=int, default=0, help="training view whose intrinsics the render uses")
    ap.add_argument("--resolution", default="336,336", help="w,h of the renders (the reference hard-wires 336x336)")
    ap.add_argument("--max-frames", type=int, default=0, help="(c): render at most this many poses, evenly spaced (0 = all)")
    ap.add_argument("--score-tol", type=float, default=1e-3, help="(b): bar on |score - reference| / |reference| (1e-3 cosine -> ~2e-3 on the ratio)")
    ap.add_argument("--psnr-min", type=float, default=40.0, help="(c): bar on the worst frame's PSNR in dB")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-gpu", action="store_true", help="host-only sections")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    d = args.method_out
    report = {"method_out": d, "sections": {}}
    ok, ran = True, 0

    # ---------------- (a) snapshots, host side
    snaps = {k: os.path.join(d, f"{k}_base.ingp") for k in ("fg", "bg")}
    sec = {}
    for k, p in snaps.items():
        if os.path.exists(p):
            try:
                sec[k] = inspect_snapshot(p)
            except Exception as e:          # noqa: BLE001
                sec[k] = {"file": p, "problems": [f"unreadable: {e}"]}
            ran += 1
            ok = ok and not sec[k]["problems"]
        else:
            sec[k] = {"skipped": f"{p} not found"}
    report["sections"]["a_snapshots"] = sec

    gpu = None
    if not args.no_gpu:
        try:
            from dream2real_amd import engine
            gpu = engine.Context(args.device)
        except Exception as e:          # noqa: BLE001
            report["gpu"] = f"unavailable: {e}"
    if gpu is not None:
        from dream2real_amd import engine
        for k, p in snaps.items():
            if os.path.exists(p):
                try:
                    tb = engine.Testbed.from_snapshot(gpu, p)
                    sec[k]["gpu_loader"] = {"ok": True, "training_views": len(tb.training_views), "dataset_scale": tb.dataset_scale,
                                            "dataset_offset": list(tb.dataset_offset), "background_color": list(tb.background_color)}
                    sec[k]["_tb"] = tb
                except Exception as e:          # noqa: BLE001
                    sec[k]["gpu_loader"] = {"ok": False, "error": str(e)}
                    ok = False

    # ---------------- inputs shared by (b)-(d)
    scores_path, batch_path, goal_path = (os.path.join(d, f) for f in ("pose_scores.txt", "pose_batch.txt", "goal_pose.txt"))
    render_dir = os.path.join(d, "cb_render")
    old_scores = np.loadtxt(scores_path) if os.path.exists(scores_path) else None
    pose_batch = np.loadtxt(batch_path).reshape(-1, 16) if os.path.exists(batch_path) else None
    frames = files = None
    if os.path.isdir(render_dir) and glob.glob(os.path.join(render_dir, "*.png")):
        frames, files = load_frames(render_dir)
    sample_res = [int(x) for x in args.sample_res.split(",")] if args.sample_res else None
    goal, norms = args.goal_caption, args.norm_caption
    if args.captions_json and args.scene:
        c = json.load(open(args.captions_json))[args.scene]
        goal = goal or c.get("goal_caption")
        norms = norms or c.get("norm_captions")

    # ---------------- (b) CLIP-only parity on the cached renders
    new_scores = None
    if gpu is None or not args.clip or old_scores is None or frames is None or not goal:
        missing = [n for n, v in (("a GPU", gpu), ("--clip", args.clip), ("pose_scores.txt", old_scores is not None), ("cb_render/*.png", frames), ("a goal caption", goal)) if not v]
        report["sections"]["b_clip"] = {"skipped": "needs " + ", ".join(missing)}
    else:
        from dream2real_amd import clip_model, engine
        from dream2real_amd.clip_scoring import build_captions, reduce_logits
        from dream2real_amd.geometry_utils import spatially_smooth_heatmap
        from dream2real_amd.tokenizer import ClipBpeTokenizer
        cfg, sd = clip_model.load_clip_safetensors(os.path.join(args.clip, "model.safetensors"))
        tok = ClipBpeTokenizer.from_files(os.path.join(args.clip, "vocab.json"), os.path.join(args.clip, "merges.txt"), context_length=cfg["ctx"])
        scorer, enc = engine.ClipScorer(gpu, cfg, sd), engine.TextEncoder(gpu, cfg, sd)
        captions, n_goal = build_captions(goal, norms, False)
        ids = tok(captions)
        text = enc.encode(np.asarray(ids[0] if isinstance(ids, tuple) else ids, np.int32))
        valid = np.nonzero(old_scores)[0]                                   # clip_scoring.py:92-94
        sec = {"frames": len(frames), "valid_poses": int(len(valid)), "captions": captions,
               "clip": {k: cfg[k] for k in ("image_size", "patch_size", "hidden_size", "num_layers", "proj")}}
        if len(frames) != len(valid):
            sec["problem"] = f"Expected {len(valid)} renders, got {len(frames)}"
            ok = False
        else:
            shapes = {f.shape for f in frames}
            assert len(shapes) == 1, f"renders of different sizes: {shapes}"
            logits = scorer.score_frames(np.stack(frames), text, rot90=True)      # clip_scoring.py:145-185
            ratio = reduce_logits(logits, n_goal, norms is not None)
            raw = np.zeros(len(old_scores), np.float32)
            raw[valid] = ratio
            cands = {"raw": raw}
            if sample_res is not None and int(np.prod(sample_res)) == len(old_scores):
                cands["smoothed"] = spatially_smooth_heatmap(raw.copy(), sample_res)
            best_name, best_err = None, np.inf
            for name, s in cands.items():
                rel = np.abs(s[valid] - old_scores[valid]) / np.maximum(np.abs(old_scores[valid]), 1e-12)
                sec[name] = {"max_rel_err": float(rel.max()), "mean_rel_err": float(rel.mean()), "spearman": spearman(s[valid], old_scores[valid]),
                             "argmax_identical": bool(int(np.argmax(s)) == int(np.argmax(old_scores)))}
                if rel.max() < best_err:
                    best_name, best_err, new_scores = name, float(rel.max()), s
            sec["matches"] = best_name
            sec["within_bar"] = bool(best_err <= args.score_tol)
            ok = ok and sec["within_bar"]
        ran += 1
        report["sections"]["b_clip"] = sec
        scorer.close(); enc.close()

    # ---------------- (c) render parity against the cached PNGs
    fg_tb = report["sections"]["a_snapshots"].get("fg", {}).get("_tb")
    bg_tb = report["sections"]["a_snapshots"].get("bg", {}).get("_tb")
    need = [n for n, v in (("a GPU", gpu), ("both snapshots loaded", fg_tb is not None and bg_tb is not None), ("pose_batch.txt", pose_batch is not None),
                           ("pose_scores.txt", old_scores is not None), ("cb_render/*.png", frames), ("--obj-pose", args.obj_pose), ("--cam-pose", args.cam_pose)) if not v]
    if need:
        report["sections"]["c_render"] = {"skipped": "needs " + ", ".join(need)}
    else:
        import types
        import torch
        from dream2real_amd import accio2ngp, combined_rendering
        W, H = (int(x) for x in args.resolution.split(","))
        valid = np.nonzero(old_scores)[0]
        pick = np.arange(len(valid)) if not args.max_frames or args.max_frames >= len(valid) else \
            np.unique(np.linspace(0, len(valid) - 1, args.max_frames).astype(int))
        task = types.SimpleNamespace(movable_obj=types.SimpleNamespace(vis_model=fg_tb, pose=torch.tensor(np.loadtxt(args.obj_pose), dtype=torch.float32)),
                                     task_bground_obj=types.SimpleNamespace(vis_model=bg_tb))
        tmp = os.path.join(d, "_validate_tmp")
        rend = combined_rendering.renderer(tmp, task, resolution=(W, H))
        cam = accio2ngp.converter(np.loadtxt(args.cam_pose).reshape(1, 4, 4).astype(np.float32))
        poses = accio2ngp.converter(pose_batch[valid[pick]].reshape(-1, 4, 4).astype(np.float32))
        got = rend.render(poses, cam, [args.view_idx], None, None, save=False)
        psnr, hist = [], np.zeros(256, np.int64)
        for g, i in zip(got, pick):
            ref = frames[i]
            if ref.shape != g.shape:
                report["sections"]["c_render"] = {"problem": f"render {g.shape} vs png {ref.shape}: pass --resolution"}
                ok = False
                break
            diff = np.abs(g.astype(np.int32) - ref.astype(np.int32))
            hist += np.bincount(diff.reshape(-1), minlength=256)
            mse = float((diff.astype(np.float64) ** 2).mean())
            psnr.append(99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse))
        else:
            tot = hist.sum()
            sec = {"frames_compared": int(len(pick)), "psnr_db_min": float(min(psnr)), "psnr_db_mean": float(np.mean(psnr)),
                   "lsb_histogram": {"0": float(hist[0] / tot), "1": float(hist[1] / tot), "2": float(hist[2] / tot), "3-7": float(hist[3:8].sum() / tot),
                                     ">=8": float(hist[8:].sum() / tot)}, "max_abs_diff": int(np.nonzero(hist)[0].max())}
            sec["within_bar"] = bool(sec["psnr_db_min"] >= args.psnr_min)
            ok = ok and sec["within_bar"]
            report["sections"]["c_render"] = sec
        ran += 1
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)

    # ---------------- (d) argmax identity
    if new_scores is None or pose_batch is None or not os.path.exists(goal_path):
        report["sections"]["d_argmax"] = {"skipped": "needs section (b), pose_batch.txt and goal_pose.txt"}
    else:
        goal_pose = np.loadtxt(goal_path).reshape(16)
        best = int(np.argmax(new_scores))
        ref_idx = int(np.argmin(np.abs(pose_batch - goal_pose[None]).max(1)))
        sec = {"argmax_here": best, "goal_pose_index_in_pose_batch": ref_idx, "identical": bool(best == ref_idx),
               "goal_pose_is_a_grid_pose": bool(np.abs(pose_batch[ref_idx] - goal_pose).max() < 1e-5)}
        if not sec["identical"]:
            order = np.argsort(-new_scores)
            sec["rank_of_reference_goal_here"] = int(np.nonzero(order == ref_idx)[0][0])
            sec["score_gap"] = float(new_scores[best] - new_scores[ref_idx])
        ok = ok and sec["identical"]
        ran += 1
        report["sections"]["d_argmax"] = sec

    for k in ("fg", "bg"):
        report["sections"]["a_snapshots"].get(k, {}).pop("_tb", None)
    report["ok"] = bool(ok and ran > 0)
    text = json.dumps(report, indent=1)
    print(text)
    if args.out:
        open(args.out, "w").write(text)
    sys.exit(2 if ran == 0 else 0 if ok else 1)


if __name__ == "__main__":
    main()
