"""TEST INFRASTRUCTURE — the oracle-side pipeline: the CPU restatement of renderer.render (background, virtual camera,
foreground render, composite: reference reconstruction/combined_rendering.py:95-155) and of the CLIP scoring of the
frames (clip_scoring.py:145-181) for a synthetic scene.  The checker in tests/ and smoke(), and the thing bench.py's
`cpu_baseline` leg times; the product never imports it."""
from __future__ import annotations

import dataclasses

import numpy as np

from oracle import clip_ref, host_ref, render_ref


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
