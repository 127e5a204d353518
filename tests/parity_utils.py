"""Shared helpers for the parity tests and smoke(): the oracle-side pipeline (checker) and
small scene/task fixtures.  The oracle is only ever the checker here."""
from __future__ import annotations

import dataclasses
import types

import numpy as np

from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
from tests.scenes import make_scene, make_task, scene_text_embeds  # noqa: F401 (re-exported)
from oracle import clip_ref, host_ref, render_ref


def seeded_text_embeds(cfg, sd, n_caps=2, seed=5):
    """Cached text embeddings: the text tower of the same random checkpoint on fixed
    synthetic input_ids (EOS = largest id last), SURVEY.md §8(d)."""
    r = np.random.Generator(np.random.PCG64(seed))
    T = min(cfg["ctx"], 12)
    ids = r.integers(2, cfg["vocab"] - 2, size=(n_caps, T))
    ids[:, 0] = cfg["vocab"] - 2
    ids[:, -1] = cfg["vocab"] - 1
    return clip_ref.text_embeds(ids, sd, cfg)


def random_unit_text_embeds(D, n_caps=2, seed=5):
    r = np.random.Generator(np.random.PCG64(seed))
    t = r.standard_normal((n_caps, D)).astype(np.float32)
    return t / np.linalg.norm(t, axis=-1, keepdims=True)


class OraclePipeline:
    """CPU restatement of renderer.render + CLIP scoring for a synthetic scene."""

    def __init__(self, scene, W, H):
        self.scene, self.W, self.H = scene, W, H
        self.fg = render_ref.OracleNerf(scene.fg)
        self.bg = render_ref.OracleNerf(scene.bg)
        self.view_bg = scene.view(W, H)
        self.view_fg = dataclasses.replace(self.view_bg, background=scene.fg_background)
        self.n_samples = 0

    def background(self, view_idx=0):
        cam = host_ref.converter(np.asarray(self.scene.cam_poses, np.float32))[view_idx]
        rgba, depth, n = render_ref.render(self.bg, self.view_bg, cam[:3])
        return rgba, depth

    def fg_camera(self, pose_world, view_idx=0):
        """3x4 matrix the reference would hand to set_nerf_camera_matrix for this candidate."""
        T1 = host_ref.converter(np.asarray(self.scene.obj_pose, np.float32)[None])          # f32 [1,4,4]
        T2 = host_ref.converter(np.asarray(pose_world, np.float32).reshape(1, 4, 4))[0]
        TC = host_ref.converter(np.asarray(self.scene.cam_poses, np.float32))[view_idx]
        return host_ref.convert_virtual_pose(T1, T2, TC)[0, :3]

    def fg_render(self, pose_world, view_idx=0):
        rgba, depth, n = render_ref.render(self.fg, self.view_fg, self.fg_camera(pose_world, view_idx))
        self.n_samples += n
        return rgba, depth

    def frames(self, poses_world, view_idx=0, bg=None):
        bg_rgba, bg_depth = bg if bg is not None else self.background(view_idx)
        out = []
        for p in np.asarray(poses_world).reshape(-1, 4, 4):
            rgba, depth = self.fg_render(p, view_idx)
            out.append(render_ref.composite(rgba, depth, bg_rgba, bg_depth))
        return np.stack(out)


def oracle_logits(frames_u8, cfg, sd, text_embeds, rot90=True):
    pv = np.stack([render_ref.clip_preprocess(f, cfg["image_size"], rot90)[0] for f in frames_u8])
    emb = clip_ref.vision_embeds(pv, sd, cfg)
    return clip_ref.logits_per_image(emb, np.asarray(text_embeds, np.float32), sd["logit_scale"]), emb


def cosine(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def run_smoke():
    """__graft_entry__.smoke(): a few candidates of the tiny configuration through the HIP
    path on cuda:0, checked against the oracle."""
    from dream2real_amd import clip_scoring, combined_rendering, engine
    import tempfile

    scene = make_scene("shopping")
    W, H = 96, 54
    cfg = CLIP_CONFIGS["vit_tiny"]
    sd = random_clip_state_dict(cfg, seed=6)
    pipe = OraclePipeline(scene, W, H)
    _, e0 = oracle_logits(pipe.frames(np.asarray(scene.obj_pose, np.float32)[None]), cfg, sd, np.zeros((1, cfg["proj"])))
    text = scene_text_embeds(e0[0])
    ctx = engine.Context(0)
    fg_tb, bg_tb = engine.Testbed(ctx, scene.fg), engine.Testbed(ctx, scene.bg)
    fg_tb.background_color = list(scene.fg_background)
    scorer = engine.ClipScorer(ctx, cfg, sd)
    task = make_task(scene, fg_tb, bg_tb)
    task.text_embeds = text
    sample_res = [3, 2, 1, 1, 1, 1]
    with tempfile.TemporaryDirectory() as td:
        rend = combined_rendering.renderer(td, task, resolution=(W, H))
        all_valid = lambda pb, tm, v: v
        best, pose_batch, scores = clip_scoring.optimise_pose_grid(
            rend, None, [0], task, td, sample_res=sample_res, phys_check=all_valid, scene_type=scene.scene_type,
            smoothing=False, scorer=scorer)
    frames = pipe.frames(pose_batch.numpy())
    lg, _ = oracle_logits(frames, cfg, sd, text)
    want = host_ref.score_logits(lg, True)
    got = scores.numpy()
    err = np.abs(got - want).max()
    tol = float((0.1 * (1.0 + np.abs(want)) / np.abs(lg[:, 1])).max())    # 1e-3 cosine per logit, propagated
    print(f"smoke: {len(want)} candidates, max |score - oracle| = {err:.2e} (tol {tol:.2e}), "
          f"oracle samples = {pipe.n_samples}, libd2r loaded = {'libd2r.so' in open('/proc/self/maps').read()}")
    assert np.isfinite(got).all() and err <= tol, (got, want)
