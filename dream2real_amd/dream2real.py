"""The caller of the render-and-score path: `ImaginationEngine.dream_best_pose` (reference dream2real.py:286-358) on
the MI355X library — physics pre-filter, renderer, `optimise_pose_grid`, and the three text files the reference
persists, in the reference's order and with its switches.

Only the part of the reference's engine that drives the path is mirrored: what comes before it (segmentation, captioning,
LLM parsing, NeRF training, mesh extraction — `build_scene_model`, `build_task_model`) and what comes after it (cost-volume
visualisation, robot execution) is outside the path.  The engine object the reference keeps
these settings on (`cfg.*`, dream2real.py:40-99) is a plain settings class here; `ctx` (an engine.Context) stands where the
reference talks to PyBullet and pyngp, and `scorer` (+ text embeddings or a text encoder and tokenizer) where it
downloads CLIP.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional, Sequence

import numpy as np

from . import clip_scoring, combined_rendering, physics_utils


@dataclasses.dataclass
class PathConfig:
    """The `cfg` fields dream_best_pose reads (reference dream2real.py:40-99, cfg.py)."""
    data_dir: str
    sample_res: Sequence[int]
    scene_type: int = 0
    render_cam_pose_idx: Sequence[int] = (0,)
    use_phys: bool = True
    lazy_phys_mods: bool = True
    embodied: bool = False
    use_cache_renders: bool = False
    use_cache_goal_pose: bool = False
    spatial_smoothing: bool = True
    physics_only: bool = False
    use_vis_pcds: bool = False
    resolution: Optional[Sequence[int]] = None         # renderer resolution (w, h); None = the reference's 336 x 336
    save_renders: bool = True                          # cb_render/cb_rgb_%04d.png per valid pose (the reference always does: clip_scoring.py:140)


def compose_checks(checks):
    """reference dream2real.py:296-302: a list of validity checks -> one check that ANDs them in order"""
    def composed_check(pose_batch, task_model, valid_so_far):
        valid_so_far = valid_so_far.clone()
        for check in checks:
            valid_so_far &= check(pose_batch, task_model, valid_so_far)
        return valid_so_far
    return composed_check


class ImaginationEngine:
    """The slice of the reference's ImaginationEngine around dream_best_pose."""

    def __init__(self, cfg: PathConfig, ctx, scorer, *, text_embeds=None, text_encoder=None, tokenizer=None, depths_gt=None):
        if cfg.use_vis_pcds:
            raise NotImplementedError("the point-cloud ablation renderer is outside the path")
        self.cfg, self.ctx, self.scorer = cfg, ctx, scorer
        self.text_embeds, self.text_encoder, self.tokenizer = text_embeds, text_encoder, tokenizer
        self.depths_gt = depths_gt                       # [L, h, w] sensor depth of the render views (dream2real.py:117-118), or None
        self.data_dir = cfg.data_dir
        self.static_phys_handles = None
        self.movable_phys_handle = None
        self.renderer = None

    def dream_best_pose(self, task_model):
        """-> (best_pose [4,4], pose_batch [N,16], pose_scores [N]) as torch tensors; writes goal_pose.txt,
        pose_batch.txt, pose_scores.txt (and best_render.png, cb_render/*.png through the path) into data_dir.
        reference dream2real.py:286-358 (its cost-volume visualisation, :360-400, is not part of the path)."""
        import torch
        cfg = self.cfg
        unsupcol_check = None
        if cfg.use_phys and not cfg.use_cache_renders:                                           # :304-323
            unsupcol_check, static_obj_handles, movable_handles = physics_utils.create_unsupcol_check(
                self.ctx, task_model, cfg.sample_res, cfg.embodied, lazy_phys_mods=cfg.lazy_phys_mods)
            self.static_phys_handles = static_obj_handles
            self.movable_phys_handle = movable_handles[0]
            # the reference composes the check with a PyBullet shutdown when it is not embodied; here the GPU shapes
            # are released the same way, after the check has run
            release = lambda pose_batch, task_model, valid_so_far: (unsupcol_check.shapes.close(), valid_so_far)[1]
            phys_check = unsupcol_check if cfg.embodied else compose_checks([unsupcol_check, release])
        else:                                                                                    # :324-326
            phys_check = lambda pose_batch, task_model, valid_so_far: torch.ones(len(pose_batch), dtype=torch.bool)

        self.renderer = combined_rendering.renderer(self.data_dir, task_model, resolution=cfg.resolution)   # :332

        if cfg.use_cache_goal_pose:                                                              # :335-341
            best_pose = torch.tensor(np.loadtxt(os.path.join(self.data_dir, "goal_pose.txt"))).float()
            pose_batch = torch.tensor(np.loadtxt(os.path.join(self.data_dir, "pose_batch.txt"))).float()
            pose_scores = torch.tensor(np.loadtxt(os.path.join(self.data_dir, "pose_scores.txt"))).float()
            return best_pose, pose_batch, pose_scores
        best_pose, pose_batch, pose_scores = clip_scoring.optimise_pose_grid(                    # :343-355
            self.renderer, self.depths_gt, list(cfg.render_cam_pose_idx), task_model, self.data_dir,
            sample_res=list(cfg.sample_res), phys_check=phys_check, use_templates=False, scene_type=cfg.scene_type,
            use_vis_pcds=cfg.use_vis_pcds, use_cache_renders=cfg.use_cache_renders, smoothing=cfg.spatial_smoothing,
            physics_only=cfg.physics_only, scorer=self.scorer, text_embeds=self.text_embeds, text_encoder=self.text_encoder,
            tokenizer=self.tokenizer, save_renders=cfg.save_renders)
        from .dist import process_rank
        if process_rank() == 0:        # pose-sharded runs: every rank holds the same result, one of them writes it
            clip_scoring.save_pose_outputs(self.data_dir, best_pose, pose_batch, pose_scores)     # :356-358
        return best_pose, pose_batch, pose_scores
