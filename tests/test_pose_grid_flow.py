"""Control flow of optimise_pose_grid (reference clip_scoring.py:71-234) with a fake renderer and
scorer: validity scatter, score ratio, smoothing, argmax/best pose, best_render.png, cached
renders, physics-only and template branches.  CPU only."""
import dataclasses
import os
import types

import numpy as np
import pytest

from dream2real_amd import clip_scoring
from oracle import host_ref


class FakeRenderer:
    """renderer.render surface: K frames whose mean encodes the pose translation."""

    def __init__(self, h=12, w=20):
        self.h, self.w, self.calls = h, w, []

    def render(self, valid_poses, render_poses, render_cam_pose_idx, depths_gt=None, movable_masks=None, save=True):
        self.calls.append((np.array(valid_poses), np.array(render_poses), list(render_cam_pose_idx), save))
        out = []
        for T in np.asarray(valid_poses):
            f = np.zeros((self.h, self.w, 3), np.uint8)
            f[..., 0] = int(np.clip((T[0, 3] + 1) * 60, 0, 255))
            f[..., 1] = int(np.clip((T[1, 3] + 1) * 60, 0, 255))
            f[: self.h // 2, :, 2] = 200
            out.append(f)
        return out


class FakeScorer:
    def __init__(self):
        self.seen = None

    def score_frames(self, frames, text_embeds, rot90=True):
        self.seen = (np.array(frames), np.array(text_embeds), rot90)
        f = np.asarray(frames, np.float32)
        g = 20 + f[..., 0].mean(axis=(1, 2)) / 10 + f[..., 1].mean(axis=(1, 2)) / 20
        cols = [g] + [np.full_like(g, 18.0 + 0.1 * c) for c in range(text_embeds.shape[0] - 1)]
        return np.stack(cols, 1).astype(np.float32)


def _task(tmp, norm=("n",)):
    import torch
    sm = types.SimpleNamespace(scene_centre=torch.tensor([0.5, 0.0, 0.035]),
                               opt_cam_poses=[torch.eye(4), torch.eye(4) * 2])
    return types.SimpleNamespace(scene_model=sm, goal_caption="g", norm_captions=list(norm) if norm else None,
                                 movable_masks=None, movable_obj=types.SimpleNamespace(pose=torch.eye(4)))


def _valid(mask):
    def check(pose_batch, task_model, valid_so_far):
        import torch
        v = valid_so_far.clone()
        v[~torch.from_numpy(mask)] = False
        return v
    return check


def test_scores_scatter_smoothing_argmax(tmp_path):
    res = [6, 5, 2, 1, 1, 1]
    N = 60
    mask = np.ones(N, bool)
    mask[[0, 7, 33]] = False
    task = _task(tmp_path)
    rend, sc = FakeRenderer(), FakeScorer()
    text = np.eye(2, 8, dtype=np.float32)
    best, poses, scores = clip_scoring.optimise_pose_grid(rend, None, [1], task, str(tmp_path), sample_res=res,
                                                          phys_check=_valid(mask), scene_type=3, scorer=sc,
                                                          text_embeds=text)
    poses, scores = poses.numpy(), scores.numpy()
    np.testing.assert_array_equal(poses, host_ref.sample_poses_grid([0.5, 0.0, 0.035], res, 3))
    # the renderer saw exactly the valid poses, y/z-flipped (converter), and the chosen view
    vp, rp, idx, save = rend.calls[0]
    np.testing.assert_array_equal(vp, host_ref.converter(poses[mask].reshape(-1, 4, 4)))
    np.testing.assert_array_equal(rp, host_ref.converter(np.eye(4)[None] * 2))
    assert idx == [1] and sc.seen[2] is True
    # expected scores through the oracle restatements
    lg = sc.score_frames(rend.render(vp, rp, idx), text)
    want = np.zeros(N, np.float32)
    want[mask] = host_ref.score_logits(lg, True)
    want = host_ref.spatially_smooth_heatmap(want, res)
    np.testing.assert_allclose(scores, want, rtol=0, atol=1e-6)
    assert (scores[~mask] == 0).all()
    bi = int(np.argmax(want))
    np.testing.assert_array_equal(best.numpy().reshape(16), poses[bi])
    assert os.path.exists(tmp_path / "best_render.png")
    from PIL import Image
    im = np.asarray(Image.open(tmp_path / "best_render.png"))
    assert im.shape == (20, 12, 3)                                   # rot90 of a 12x20 frame


def test_no_norm_caption_and_no_smoothing(tmp_path):
    task = _task(tmp_path, norm=None)
    rend, sc = FakeRenderer(), FakeScorer()
    best, poses, scores = clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=[3, 3, 1, 1, 1, 1],
                                                          phys_check=lambda p, t, v: v, scene_type=0, smoothing=False,
                                                          scorer=sc, text_embeds=np.ones((1, 4), np.float32))
    lg = sc.score_frames(rend.render(*rend.calls[0][:3]), np.ones((1, 4), np.float32))
    np.testing.assert_allclose(scores.numpy(), lg[:, 0], rtol=0, atol=1e-6)


def test_cached_renders_roundtrip(tmp_path):
    """use_cache_renders (clip_scoring.py:89-104): validity from pose_scores.txt, frames from cb_render/."""
    from PIL import Image
    res = [4, 3, 1, 1, 1, 1]
    mask = np.ones(12, bool)
    mask[[2, 9]] = False
    task = _task(tmp_path)
    rend, sc = FakeRenderer(), FakeScorer()
    text = np.eye(2, 8, dtype=np.float32)
    _, poses, scores = clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                                       phys_check=_valid(mask), scene_type=3, scorer=sc, text_embeds=text)
    np.savetxt(tmp_path / "pose_scores.txt", scores.numpy())           # what dream2real.py:358 writes
    os.makedirs(tmp_path / "cb_render")
    for i, f in enumerate(rend.render(*rend.calls[0][:3])):
        Image.fromarray(f).save(tmp_path / "cb_render" / f"cb_rgb_{i:04d}.png")
    rend2 = FakeRenderer()
    _, _, scores2 = clip_scoring.optimise_pose_grid(rend2, None, [0], task, str(tmp_path), sample_res=res,
                                                    phys_check=None, scene_type=3, use_cache_renders=True, scorer=sc,
                                                    text_embeds=text)
    assert not rend2.calls
    np.testing.assert_allclose(scores2.numpy(), scores.numpy(), rtol=0, atol=1e-6)
    os.remove(tmp_path / "cb_render" / "cb_rgb_0000.png")
    with pytest.raises(AssertionError):
        clip_scoring.optimise_pose_grid(rend2, None, [0], task, str(tmp_path), sample_res=res, scene_type=3,
                                        use_cache_renders=True, scorer=sc, text_embeds=text)


def test_physics_only_templates_and_errors(tmp_path):
    task = _task(tmp_path)
    rend, sc = FakeRenderer(), FakeScorer()
    res = [3, 2, 1, 1, 1, 1]
    best, poses, scores = clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                                          phys_check=lambda p, t, v: v, scene_type=3,
                                                          physics_only=True, scorer=sc)
    assert not rend.calls and (scores.numpy() == 1).all() and tuple(best.shape) == (4, 4)
    # templates: 9 goal + 9 normalising captions -> 18 text embeddings
    text = np.eye(18, 32, dtype=np.float32)
    _, _, s2 = clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                               phys_check=lambda p, t, v: v, scene_type=3, use_templates=True,
                                               smoothing=False, scorer=sc, text_embeds=text)
    lg = sc.score_frames(rend.render(*rend.calls[-1][:3]), text)
    np.testing.assert_allclose(s2.numpy(), host_ref.score_logits_templates(lg, 9, True), rtol=1e-6)
    with pytest.raises(AssertionError):          # wrong number of text embeddings
        clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                        phys_check=lambda p, t, v: v, scene_type=3, scorer=sc,
                                        text_embeds=np.eye(3, 8, dtype=np.float32))
    with pytest.raises(Exception):               # no pose survives the pre-render checks
        clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                        phys_check=lambda p, t, v: v & False, scene_type=3, scorer=sc, text_embeds=text)
    with pytest.raises(ValueError):
        clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                        phys_check=lambda p, t, v: v, scene_type=3, text_embeds=text)
    with pytest.raises(NotImplementedError):
        clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=res,
                                        phys_check=lambda p, t, v: v, scene_type=3, use_vis_pcds=True, scorer=sc,
                                        text_embeds=text)


def test_text_encoder_hook(tmp_path):
    """captions -> tokenizer -> text encoder happens once and feeds the scorer."""
    task = _task(tmp_path)
    rend, sc = FakeRenderer(), FakeScorer()
    calls = []

    class Enc:
        def encode(self, ids):
            calls.append(np.array(ids))
            return np.eye(ids.shape[0], 8, dtype=np.float32)

    tok = lambda caps: [[len(c), 7, 9] for c in caps]
    clip_scoring.optimise_pose_grid(rend, None, [0], task, str(tmp_path), sample_res=[2, 2, 1, 1, 1, 1],
                                    phys_check=lambda p, t, v: v, scene_type=3, scorer=sc, text_encoder=Enc(), tokenizer=tok)
    assert len(calls) == 1 and calls[0].shape == (2, 3) and calls[0][0, 0] == 1       # "g", "n"
    np.testing.assert_array_equal(sc.seen[1], np.eye(2, 8, dtype=np.float32))


@pytest.mark.gpu
def test_dream_best_pose_flow_with_mesh_file_physics(tmp_path):
    """The caller of the path (reference dream2real.py:286-358) end to end on the GPU: physics pre-filter from the
    objects' .obj files -> renderer -> optimise_pose_grid -> goal_pose / pose_batch / pose_scores.txt, then the cached
    goal pose read back.  Invalid poses per the oracle's restatement of unsupcol_check score exactly 0, the valid ones
    match the oracle pipeline, one PNG per valid pose."""
    import torch
    from dream2real_amd import dream2real, engine, physics_utils
    from dream2real_amd.clip_model import CLIP_CONFIGS, random_clip_state_dict
    from oracle import host_ref, phys_ref
    from tests.parity_utils import OraclePipeline, oracle_logits, scene_text_embeds
    from tests.scenes import make_scene, make_task
    from tests.test_physics import box, icosphere
    scene = make_scene("shopping")
    c = np.asarray(scene.scene_centre, np.float64)
    # physics shapes in world coordinates, as get_phys_models would have written them: the table slab and the three
    # blobs of the synthetic background in ONE file (lazy_phys_mods: the merged background object), the apple in another
    table = box([c[0] - 0.6, c[1] - 0.6, -0.06], [c[0] + 0.6, c[1] + 0.6, 0.0])
    blobs = [icosphere(c + np.array([dx, dy, r - 0.035]), r, 40, k) for k, (dx, dy, r) in enumerate(((0.12, -0.10, 0.05), (-0.15, 0.08, 0.04), (0.05, 0.15, 0.045)))]
    apple = icosphere(np.asarray(scene.obj_pose)[:3, 3], 0.03, 40, 9)
    bg_path, mov_path = str(tmp_path / "bground.obj"), str(tmp_path / "movable.obj")
    with open(bg_path, "w") as f:
        for k, h in enumerate([table] + blobs):
            f.write(f"o part_{k}\n" + "".join(f"v {x:.9g} {y:.9g} {z:.9g}\n" for x, y, z in h))
            base = sum(len(q) for q in ([table] + blobs)[:k])
            f.write("".join(f"f {base + i + 1} {base + i + 2} {base + i + 3}\n" for i in range(len(h) - 2)))      # any faces that reference every vertex
    open(mov_path, "w").write("".join(f"v {x:.9g} {y:.9g} {z:.9g}\n" for x, y, z in apple))
    assert [len(h) for h in physics_utils.hulls_from_obj(bg_path)] == [8, 40, 40, 40] and len(physics_utils.hulls_from_obj(mov_path)[0]) == 40

    ctx = engine.Context(0)
    fg, bg = engine.Testbed(ctx, scene.fg), engine.Testbed(ctx, scene.bg)
    fg.background_color = list(scene.fg_background)
    cfg_clip = CLIP_CONFIGS["vit_tiny"]
    sd = random_clip_state_dict(cfg_clip, seed=6)
    sc = engine.ClipScorer(ctx, cfg_clip, sd)
    W, H = 96, 54
    pipe = OraclePipeline(scene, W, H)
    _, e0 = oracle_logits(pipe.frames(np.asarray(scene.obj_pose, np.float32)[None]), cfg_clip, sd, np.zeros((1, cfg_clip["proj"])))
    task = make_task(scene, fg, bg)
    task.text_embeds = scene_text_embeds(e0[0])
    task.movable_obj.phys_model, task.task_bground_obj.phys_model = mov_path, bg_path
    sample_res = [7, 6, 2, 1, 1, 1]
    d = str(tmp_path / "run")
    os.makedirs(d)
    cfg = dream2real.PathConfig(data_dir=d, sample_res=sample_res, scene_type=scene.scene_type, resolution=(W, H), spatial_smoothing=False)
    eng = dream2real.ImaginationEngine(cfg, ctx, sc)
    best, pose_batch, scores = eng.dream_best_pose(task)
    assert len(eng.static_phys_handles) == 1 and len(eng.static_phys_handles[0]) == 4 and len(eng.movable_phys_handle) == 1
    # oracle: physics mask, then render + score the valid poses
    valid = phys_ref.unsupcol_check(pose_batch.numpy(), np.asarray(scene.obj_pose, np.float32), [apple], [table] + blobs, sample_res,
                                    np.ones(84, bool), float(c[2]), margin=physics_utils.PYBULLET_MESH_MARGIN)
    got = scores.numpy()
    assert 5 < valid.sum() < 80, valid.sum()
    # equal to the oracle's mask except where a hull pair lies within rounding of the contact distance (the oracle's answer
    # for a margin 5e-6 m smaller or larger)
    m0 = physics_utils.PYBULLET_MESH_MARGIN
    near = [phys_ref.unsupcol_check(pose_batch.numpy(), np.asarray(scene.obj_pose, np.float32), [apple], [table] + blobs, sample_res,
                                    np.ones(84, bool), float(c[2]), margin=m) for m in (m0 - 5e-6, m0 + 5e-6)]
    assert (((got != 0) == valid) | ((got != 0) == near[0]) | ((got != 0) == near[1])).all()
    both = valid & (got != 0)
    frames = pipe.frames(pose_batch.numpy()[both])
    lg, _ = oracle_logits(frames, cfg_clip, sd, task.text_embeds)
    want = host_ref.score_logits(lg, True)
    from tests.parity_utils import logit_bar
    # the per-logit bar (1e-3 cosine, dimension-corrected for the D = 64 test model) propagated through goal / norm
    tol = float((100.0 * logit_bar(cfg_clip) * (1.0 + np.abs(want)) / np.abs(lg[:, 1])).max())
    print(f"[parity] dream_best_pose flow (vit_tiny): max |score - oracle| = {np.abs(got[both] - want).max():.2e} (propagated bar {tol:.2e})")
    np.testing.assert_allclose(got[both], want, rtol=0, atol=tol)
    assert len(os.listdir(os.path.join(d, "cb_render"))) == int((got != 0).sum())
    for name in ("goal_pose.txt", "pose_batch.txt", "pose_scores.txt", "best_render.png"):
        assert os.path.exists(os.path.join(d, name)), name
    np.testing.assert_allclose(np.loadtxt(os.path.join(d, "pose_scores.txt")), got, rtol=1e-6)
    assert got[int(np.argmax(got))] > 0 and np.allclose(best.numpy().reshape(16), pose_batch.numpy()[int(np.argmax(got))])
    # the cached goal pose (use_cache_goal_pose, reference :335-341)
    eng2 = dream2real.ImaginationEngine(dataclasses.replace(cfg, use_cache_goal_pose=True), ctx, sc)
    b2, p2, s2 = eng2.dream_best_pose(task)
    np.testing.assert_allclose(b2.numpy(), best.numpy(), rtol=1e-6)
    np.testing.assert_allclose(s2.numpy(), got, rtol=1e-6)
    # physics off: every pose is rendered (reference :324-326)
    d3 = str(tmp_path / "run3")
    os.makedirs(d3)
    eng3 = dream2real.ImaginationEngine(dataclasses.replace(cfg, data_dir=d3, use_phys=False, sample_res=[3, 2, 1, 1, 1, 1]), ctx, sc)
    _, _, s3 = eng3.dream_best_pose(task)
    assert (s3.numpy() != 0).all() and len(os.listdir(os.path.join(d3, "cb_render"))) == 6
    sc.close(); fg.close(); bg.close(); ctx.close()
