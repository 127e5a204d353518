"""Pose-batch in, scores out: the render-and-score driver on MI355X.

Mirrors reference clip_scoring.py:71-234 (`optimise_pose_grid`): same positional arguments,
same return triple (best_pose 4x4, pose_batch [N,16], pose_scores [N]) as torch tensors, same
failure on zero valid poses.  Differences forced by the environment are keyword-only extras:
`scorer` (a dream2real_amd.engine.ClipScorer holding the CLIP weights on the GPU) and either
`text_embeds` (cached, L2-normalised [C,D]) or `text_encoder` + `tokenizer` (an
engine.TextEncoder and a callable captions -> int32 ids [C,T], e.g. tokenizer.ClipBpeTokenizer
built from the checkpoint's vocab.json / merges.txt; the embeddings are then computed
once on the GPU — the reference re-tokenises and re-encodes the captions every batch).
"""
from __future__ import annotations

import os

import numpy as np

from . import accio2ngp
from .geometry_utils import spatially_smooth_heatmap
from .obj_pose_opt import sample_poses_grid
from .virtual_cam_pose_sample import get_virtual_cam_poses

CLIP_RES = 336

# Prompt templates of the (cold) use_templates branch; the strings are data and must equal the
# reference's (clip_text_templates.py:1-11) for the branch to score the same prompts.  The
# caller passes use_templates=False (dream2real.py:350).
CLIP_TEMPLATES = tuple(p + "{}" for p in (
    "", "a photo of ", "a bad photo of ", "a good photo of ", "a low resolution photo of ",
    "a cropped photo of ", "a bright photo of ", "a dark photo of ", "a painting of "))


def reduce_logits(all_logits: np.ndarray, n_captions_goal: int, has_norm: bool) -> np.ndarray:
    """[K,C] logits_per_image -> [K] score: goal / mean(normalising), averaged over templates
    when used (reference clip_scoring.py:187-203)."""
    a = np.asarray(all_logits, np.float32)
    if not has_norm:
        return a.mean(axis=1, dtype=np.float32) if a.shape[1] > 1 else a[:, 0].copy()
    goal = a[:, :n_captions_goal].mean(axis=1, dtype=np.float32) if n_captions_goal > 1 else a[:, 0]
    norm = a[:, n_captions_goal:].mean(axis=1, dtype=np.float32)
    return (goal / norm).astype(np.float32)


def build_captions(goal_caption, norm_captions, use_templates):
    """reference clip_scoring.py:153-163 -> (captions, number of goal captions)."""
    if use_templates:
        captions = [t.format(goal_caption) for t in CLIP_TEMPLATES]
        n_goal = len(captions)
        if norm_captions is not None:
            for c in norm_captions:
                captions += [t.format(c) for t in CLIP_TEMPLATES]
        return captions, n_goal
    return ([goal_caption] if norm_captions is None else [goal_caption] + list(norm_captions)), 1


def save_pose_outputs(data_dir, best_pose, pose_batch, pose_scores):
    """What the reference's caller persists after optimise_pose_grid (dream2real.py:356-358):
    goal_pose.txt, pose_batch.txt, pose_scores.txt in np.savetxt's text format — the files a later
    run with use_cache_goal_pose / use_cache_renders reads back."""
    def arr(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    np.savetxt(os.path.join(data_dir, "goal_pose.txt"), arr(best_pose))
    np.savetxt(os.path.join(data_dir, "pose_batch.txt"), arr(pose_batch))
    np.savetxt(os.path.join(data_dir, "pose_scores.txt"), arr(pose_scores))


def optimise_pose_grid(renderer, depths_gt, render_cam_pose_idx, task_model, data_dir, sample_res=None,
                       phys_check=None, use_templates=False, scene_type=0, use_vis_pcds=False,
                       use_cache_renders=False, smoothing=True, physics_only=False, *, scorer=None,
                       text_embeds=None, text_encoder=None, tokenizer=None, show=False, save_renders=True):
    import torch

    if sample_res is None:
        sample_res = [40, 40, 1, 1, 1, 1]
    pose_batch = sample_poses_grid(task_model, sample_res, scene_type=scene_type)
    N = pose_batch.shape[0]

    if use_cache_renders:
        old = np.loadtxt(os.path.join(data_dir, "pose_scores.txt"))
        valid_idxs = np.nonzero(old)[0]
        valid_poses = pose_batch[valid_idxs]
        from PIL import Image
        render_dir = os.path.join(data_dir, "cb_render")
        renders = [np.asarray(Image.open(os.path.join(render_dir, f)).convert("RGB"))
                   for f in sorted(os.listdir(render_dir))]
        assert len(renders) == valid_poses.shape[0], \
            f"Expected {valid_poses.shape[0]} renders, got {len(renders)}. Try running without use_cache_renders."
    else:
        valid_so_far = torch.ones(N).bool()
        is_valid = phys_check(torch.from_numpy(pose_batch), task_model, valid_so_far)
        valid_idxs = np.nonzero(np.asarray(is_valid, bool))[0]
        valid_poses = pose_batch[valid_idxs]
        if valid_idxs.shape[0] == 0:
            print("No poses passed pre-render checks. Exiting.")
            raise Exception
        if physics_only:
            best = int(torch.randint(valid_idxs.shape[0], (1,)).item())
            return torch.from_numpy(valid_poses[best].reshape(4, 4).copy()), torch.from_numpy(pose_batch), torch.ones(N)
        if use_vis_pcds:
            raise NotImplementedError("the point-cloud ablation renderer is outside the path")
        render_poses = get_virtual_cam_poses(task_model, render_cam_pose_idx)
        render_poses_ngp = accio2ngp.converter(render_poses)
        valid_poses_ngp = accio2ngp.converter(valid_poses.reshape(-1, 4, 4))
        renders = renderer.render(valid_poses_ngp, render_poses_ngp, render_cam_pose_idx, depths_gt,
                                  getattr(task_model, "movable_masks", None), save=save_renders)   # reference :135-140: save=True

    if hasattr(task_model, "free_visual_models"):
        pass    # the reference frees the NeRFs here to make room for CLIP; 288 GB makes that unnecessary

    if scorer is None:
        raise ValueError("optimise_pose_grid needs scorer=ClipScorer(...): CLIP weights cannot be downloaded here")
    captions, n_goal = build_captions(task_model.goal_caption, task_model.norm_captions, use_templates)
    if text_embeds is None:
        text_embeds = getattr(task_model, "text_embeds", None)
    if text_embeds is None and text_encoder is not None and tokenizer is not None:
        ids = tokenizer(captions)                      # ids, or (ids, attention_mask) like tokenizer.ClipBpeTokenizer
        if isinstance(ids, tuple):
            ids = ids[0]
        text_embeds = text_encoder.encode(np.asarray(ids, np.int32))   # once per task
    if text_embeds is None:
        raise ValueError("text embeddings are required: pass text_embeds=, or text_encoder= and tokenizer=")
    text_embeds = np.asarray(text_embeds, np.float32)
    assert text_embeds.shape[0] == len(captions), "one text embedding per caption"

    # rot90 (reference :145) + processor + vision tower + logits run on the GPU in one call
    all_logits = scorer.score_frames(np.stack(renders), text_embeds, rot90=True)
    logits = reduce_logits(all_logits, n_goal, task_model.norm_captions is not None)

    pose_scores = np.zeros(N, np.float32)
    pose_scores[valid_idxs] = logits
    render_idxs = np.zeros(N, np.int64)
    render_idxs[valid_idxs] = np.arange(valid_idxs.shape[0])
    if smoothing:
        pose_scores = spatially_smooth_heatmap(pose_scores, sample_res)
    best_pose_idx = int(np.argmax(pose_scores))
    best_render = np.rot90(renders[render_idxs[best_pose_idx]], k=1, axes=(0, 1))
    best_pose = valid_poses[render_idxs[best_pose_idx]]
    from PIL import Image
    img = Image.fromarray(np.ascontiguousarray(best_render))
    img.save(os.path.join(data_dir, "best_render.png"))
    if show:
        img.show()
    if hasattr(renderer, "wait_saved"):
        renderer.wait_saved()          # cb_render/*.png are complete when the call returns, as in the reference
    return (torch.from_numpy(best_pose.reshape(4, 4).copy()), torch.from_numpy(pose_batch),
            torch.from_numpy(pose_scores))
