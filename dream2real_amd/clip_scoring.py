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
    from . import _lib

    def arr(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    # d2r_savetxt: np.savetxt's default text format, byte for byte, without the per-row Python loop (1.1 M numbers for the
    # reference's 70 000-pose grid)
    _lib.savetxt(os.path.join(data_dir, "goal_pose.txt"), arr(best_pose))
    _lib.savetxt(os.path.join(data_dir, "pose_batch.txt"), arr(pose_batch))
    _lib.savetxt(os.path.join(data_dir, "pose_scores.txt"), arr(pose_scores))


LAST_TIMINGS: dict = {}      # seconds per stage of the last optimise_pose_grid call (bench.py --api reports them)

CACHE_READ_CHUNK = 2048      # cached renders are read and scored this many at a time (host memory stays bounded)


def _cached_render_reader(render_dir):
    """The frames `use_cache_renders` scores (reference clip_scoring.py:95-104): every file of cb_render/ in SORTED NAME
    order — which is index order up to cb_rgb_9999.png and lexical order beyond it, as in the reference.
    -> (n, read(j0, j1) -> uint8 [j1-j0,H,W,3])."""
    import re
    from . import _lib
    names = sorted(os.listdir(render_dir))
    m = [re.fullmatch(r"cb_rgb_(\d{4,})\.png", f) for f in names]
    if names and all(m):
        idx = np.array([int(x.group(1)) for x in m], np.uint32)
        size = _lib.png_size(os.path.join(render_dir, names[0]))
        return len(names), lambda j0, j1: _lib.png_read_batch(render_dir, indices=idx[j0:j1], size=size)

    def read_pil(j0, j1):          # other file names: whatever image files a user dropped there
        from PIL import Image
        return np.stack([np.asarray(Image.open(os.path.join(render_dir, f)).convert("RGB")) for f in names[j0:j1]])
    return len(names), read_pil


def optimise_pose_grid(renderer, depths_gt, render_cam_pose_idx, task_model, data_dir, sample_res=None,
                       phys_check=None, use_templates=False, scene_type=0, use_vis_pcds=False,
                       use_cache_renders=False, smoothing=True, physics_only=False, *, scorer=None,
                       text_embeds=None, text_encoder=None, tokenizer=None, show=False, save_renders=True, shard=None):
    """reference clip_scoring.py:71-234.  With this package's `renderer` and an engine.ClipScorer the valid poses go
    through ONE fused library call per render view (d2r_render_score_host: render -> composite -> CLIP -> logits on the
    GPU; only poses go in and logits come out, `save_renders` streams cb_render/*.png from the library's worker threads),
    in the chunks the library picks; any other renderer / scorer takes the reference's two-step route (`render`, then
    `score_frames`).  Under a launcher with WORLD_SIZE > 1 (or an explicit `shard`, a dist.PoseShard) the valid poses are
    split in contiguous blocks over the ranks, logits are all-gathered ONCE, and ratio / scatter / smoothing / argmax run
    identically on every rank; rank 0 writes best_render.png."""
    import time
    import torch
    from . import _lib

    LAST_TIMINGS.clear()
    t_mark = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        LAST_TIMINGS[name] = LAST_TIMINGS.get(name, 0.0) + now - t_mark[0]
        t_mark[0] = now

    if sample_res is None:
        sample_res = [40, 40, 1, 1, 1, 1]
    pose_batch = sample_poses_grid(task_model, sample_res, scene_type=scene_type)
    N = pose_batch.shape[0]
    lap("sample_poses_grid")

    def text_embeddings():
        if scorer is None:
            raise ValueError("optimise_pose_grid needs scorer=ClipScorer(...): CLIP weights cannot be downloaded here")
        captions, n_goal = build_captions(task_model.goal_caption, task_model.norm_captions, use_templates)
        te = text_embeds
        if te is None:
            te = getattr(task_model, "text_embeds", None)
        if te is None and text_encoder is not None and tokenizer is not None:
            ids = tokenizer(captions)                      # ids, or (ids, attention_mask) like tokenizer.ClipBpeTokenizer
            if isinstance(ids, tuple):
                ids = ids[0]
            te = text_encoder.encode(np.asarray(ids, np.int32))   # once per task
        if te is None:
            raise ValueError("text embeddings are required: pass text_embeds=, or text_encoder= and tokenizer=")
        te = np.asarray(te, np.float32)
        assert te.shape[0] == len(captions), "one text embedding per caption"
        return te, n_goal

    fetch_render = None            # render index -> uint8 [H,W,3] frame, for best_render.png
    from .dist import process_rank
    rank, world = 0, 1
    if use_cache_renders:
        old = np.loadtxt(os.path.join(data_dir, "pose_scores.txt"))
        valid_idxs = np.nonzero(old)[0]
        valid_poses = pose_batch[valid_idxs]
        render_dir = os.path.join(data_dir, "cb_render")
        n_files, read = _cached_render_reader(render_dir)
        assert n_files == valid_poses.shape[0], \
            f"Expected {valid_poses.shape[0]} renders, got {n_files}. Try running without use_cache_renders."
        te, n_goal = text_embeddings()
        # rot90 (reference :145) + processor + vision tower + logits run on the GPU, a chunk of files at a time
        parts = [scorer.score_frames(read(j0, min(n_files, j0 + CACHE_READ_CHUNK)), te, rot90=True)
                 for j0 in range(0, n_files, CACHE_READ_CHUNK)]
        all_logits = np.concatenate(parts, 0)
        fetch_render = lambda j: read(j, j + 1)[0]
    else:
        valid_so_far = torch.ones(N).bool()
        is_valid = phys_check(torch.from_numpy(pose_batch), task_model, valid_so_far)
        valid_idxs = np.nonzero(np.asarray(is_valid, bool))[0]
        valid_poses = pose_batch[valid_idxs]
        lap("phys_check")
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
        K = valid_poses_ngp.shape[0]
        te, n_goal = text_embeddings()
        lap("convert_poses_and_text")
        if hasattr(task_model, "free_visual_models"):
            pass    # the reference frees the NeRFs here to make room for CLIP; 288 GB makes that unnecessary
        L = len(render_cam_pose_idx)
        fused = hasattr(renderer, "render_score") and hasattr(scorer, "h") and L >= 1
        if shard is None and fused:
            from .dist import PoseShard
            shard = PoseShard.from_env(renderer.fg_obj.vis_model.ctx)        # None outside a multi-rank launcher
        if shard is not None:
            rank, world = shard.rank, shard.world
        masks = getattr(task_model, "movable_masks", None)
        if fused:
            lo, hi = (0, K) if shard is None else shard.range(K)
            if shard is not None and save_renders:            # rank 0 deletes the old renders, then every rank writes its block
                if rank == 0:
                    renderer._clear_renders()
                shard.barrier()
            local = renderer.render_score(valid_poses_ngp[lo:hi], render_poses_ngp, render_cam_pose_idx, scorer, te,
                                          depths_gt, masks, save=save_renders, first_index=lo, clear=shard is None)
            lap("render_score")
            if shard is None:
                all_logits = local
            else:          # view-major [L * (hi - lo), C] per rank -> one gather per view -> view-major [L * K, C], the reference's frame order
                per_view = np.asarray(local).reshape(L, hi - lo, -1)
                all_logits = np.concatenate([shard.gather(per_view[v], K) for v in range(L)], 0)
            lap("gather")
            # (a multi-view call yields L * K logits for K valid poses: the scatter below then fails exactly as the reference's
            # `pose_scores[valid_idxs] = logits` (clip_scoring.py:205-206) does — every shipped config renders one view)
            fetch_render = lambda j: renderer.render_one(valid_poses_ngp[j])
        else:
            if world > 1:
                raise NotImplementedError("pose sharding needs this package's renderer and an engine.ClipScorer")
            renders = renderer.render(valid_poses_ngp, render_poses_ngp, render_cam_pose_idx, depths_gt, masks,
                                      save=save_renders)                                        # reference :135-140: save=True
            # rot90 (reference :145) + processor + vision tower + logits run on the GPU in one call
            all_logits = scorer.score_frames(np.stack(renders), te, rot90=True)
            fetch_render = lambda j: renders[j]

    logits = reduce_logits(all_logits, n_goal, task_model.norm_captions is not None)

    pose_scores = np.zeros(N, np.float32)
    pose_scores[valid_idxs] = logits
    render_idxs = np.zeros(N, np.int64)
    render_idxs[valid_idxs] = np.arange(valid_idxs.shape[0])
    if smoothing:
        pose_scores = spatially_smooth_heatmap(pose_scores, sample_res)
    best_pose_idx = int(np.argmax(pose_scores))
    best_pose = valid_poses[render_idxs[best_pose_idx]]
    lap("reduce_scatter_smooth_argmax")
    # one writer under a launcher whichever route was taken (the cached-render and two-step routes never set `rank`)
    if rank == 0 and (shard is not None or process_rank() == 0):
        best_render = np.rot90(fetch_render(int(render_idxs[best_pose_idx])), k=1, axes=(0, 1))
        _lib.png_write(np.ascontiguousarray(best_render), os.path.join(data_dir, "best_render.png"))
        if show:
            from PIL import Image
            Image.fromarray(np.ascontiguousarray(best_render)).show()
    if hasattr(renderer, "wait_saved"):
        renderer.wait_saved()          # cb_render/*.png are complete when the call returns, as in the reference
    if shard is not None:
        shard.barrier()
    lap("best_render_png")
    return (torch.from_numpy(best_pose.reshape(4, 4).copy()), torch.from_numpy(pose_batch),
            torch.from_numpy(pose_scores))
