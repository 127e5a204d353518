"""Pose sharding + the single all-gather, world_size 2 over gloo on CPU (the N>1 path of
bench.py uses the same functions with the nccl/RCCL backend)."""
import os
import subprocess
import sys

import numpy as np

from dream2real_amd import dist as d2r_dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["D2R_REPO"])
from dream2real_amd import dist as dd
rank, world, _ = dd.init_from_env("gloo")
rng = np.random.default_rng(0)
res = [5, 3, 2, 1, 1, 1]; N = 30
poses = rng.standard_normal((N, 16)).astype(np.float32)
valid = np.ones(N, bool); valid[[3, 17]] = False
def score_fn(p):      # deterministic fake scorer: logits depend only on the pose
    a = torch.from_numpy(p)
    return torch.stack([20 + a[:, 3], 18 + 0.5 * a[:, 7]], 1)
best, scores = dd.score_sharded(poses, score_fn, res, True, rank=rank, world=world, is_valid=valid)
np.save(os.path.join(os.environ["D2R_OUT"], f"scores_{rank}.npy"), scores)
open(os.path.join(os.environ["D2R_OUT"], f"best_{rank}.txt"), "w").write(str(best))
# the bench's gather object (torch fallback of d2r_allgather_scores; ragged shards: 7 rows over 2 ranks)
assert dd.init_comm(None, rank, world) is False          # no GPU here: every rank agrees on the fallback
g = dd.ShardGather(None, 7, 2, rank, world, "cpu", False)
g.local[: g.hi - g.lo] = torch.arange(g.lo, g.hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 10.0])
np.save(os.path.join(os.environ["D2R_OUT"], f"gather_{rank}.npy"), g.gather())
'''


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 4096, 131072, 13):
        for w in (1, 2, 3, 8):
            r = [d2r_dist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, D2R_REPO=REPO, D2R_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s0, s1 = np.load(tmp_path / "scores_0.npy"), np.load(tmp_path / "scores_1.npy")
    np.testing.assert_array_equal(s0, s1)                              # identical on every rank
    assert (tmp_path / "best_0.txt").read_text() == (tmp_path / "best_1.txt").read_text()
    # single-process answer
    rng = np.random.default_rng(0)
    poses = rng.standard_normal((30, 16)).astype(np.float32)
    valid = np.ones(30, bool)
    valid[[3, 17]] = False
    fn = lambda p: torch.stack([20 + torch.from_numpy(p)[:, 3], 18 + 0.5 * torch.from_numpy(p)[:, 7]], 1)
    best, want = d2r_dist.score_sharded(poses, fn, [5, 3, 2, 1, 1, 1], True, is_valid=valid)
    np.testing.assert_array_equal(s0, want)
    assert int((tmp_path / "best_0.txt").read_text()) == best
    assert want[3] == 0 and want[17] == 0
    want_g = np.arange(7, dtype=np.float32)[:, None] * np.array([1.0, 10.0], np.float32)
    np.testing.assert_array_equal(np.load(tmp_path / "gather_0.npy"), want_g)
    np.testing.assert_array_equal(np.load(tmp_path / "gather_1.npy"), want_g)


def test_stuck_rendezvous_is_reported_not_hung(tmp_path):
    """bench.py under a launcher whose other rank never arrives: the watchdog prints ONE JSON line naming the stage,
    the rank and the HSA_* / NCCL_* / MASTER_* environment and exits non-zero instead of waiting in the rendezvous."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               D2R_WATCHDOG_SCALE="0.02", NCCL_DEBUG="WARN", D2R_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-collective"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=repo)
    assert r.returncode == 3, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "no progress" in d["collective_error"] and d["stage"].startswith("rendezvous")
    assert d["rank"] == 0 and d["world"] == 2 and d["env"]["MASTER_PORT"] == str(port) and d["env"]["NCCL_DEBUG"] == "WARN"
    assert d["machine"] and "torch" in d


API_WORKER = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.environ["D2R_REPO"])
from dream2real_amd import clip_scoring, dist as dd
rank, world, _ = dd.init_from_env("gloo")
out = os.environ["D2R_OUT"]

class FusedRenderer:                      # the surface optimise_pose_grid's fused branch uses, logits a function of the pose
    def __init__(self, root):
        self.fg_obj = types.SimpleNamespace(vis_model=types.SimpleNamespace(ctx=None))
        self.out_render_path = os.path.join(root, "cb_render")
        self.calls = []
    def _clear_renders(self):
        import shutil
        shutil.rmtree(self.out_render_path, ignore_errors=True)
        os.makedirs(self.out_render_path)
    def render_score(self, valid_poses, render_poses, idx, scorer, text, depths_gt=None, movable_masks=None, save=True, first_index=0, clear=True, return_frames=False):
        self.calls.append((len(valid_poses), first_index, clear))
        if save and clear:
            self._clear_renders()
        p = np.asarray(valid_poses, np.float32).reshape(-1, 16)
        if save:
            for i in range(len(p)):
                open(os.path.join(self.out_render_path, f"cb_rgb_{first_index + i:04d}.png"), "w").write(str(rank))
        return np.stack([20 + p[:, 3] * 3 + p[:, 7], 18 + 0.1 * p[:, 11]], 1).astype(np.float32)
    def render_one(self, pose):
        return np.zeros((4, 6, 3), np.uint8)

sm = types.SimpleNamespace(scene_centre=torch.tensor([0.5, 0.0, 0.035]), opt_cam_poses=[torch.eye(4)])
task = types.SimpleNamespace(scene_model=sm, goal_caption="g", norm_captions=["n"], movable_masks=None,
                             movable_obj=types.SimpleNamespace(pose=torch.eye(4)))
mask = np.ones(70, bool); mask[[1, 8, 40, 41, 69]] = False
def phys(pose_batch, tm, valid):
    return valid & torch.from_numpy(mask)
rend = FusedRenderer(out)
best, poses, scores = clip_scoring.optimise_pose_grid(rend, None, [0], task, out, sample_res=[7, 5, 2, 1, 1, 1], phys_check=phys, scene_type=3,
                                                      scorer=types.SimpleNamespace(h=1), text_embeds=np.eye(2, 8, dtype=np.float32))
np.save(os.path.join(out, f"api_scores_{rank}.npy"), scores.numpy())
np.save(os.path.join(out, f"api_best_{rank}.npy"), best.numpy())
open(os.path.join(out, f"api_calls_{rank}.txt"), "w").write(repr(rend.calls))
'''


def test_optimise_pose_grid_shards_inside_the_api(tmp_path):
    """optimise_pose_grid under a 2-rank launcher (gloo, CPU; fake fused renderer): every rank renders + scores its
    contiguous block of the VALID poses with the right first file index, logits are gathered once, and scores / best
    pose / files equal the single-process run on every rank; rank 0 alone writes best_render.png."""
    import ast
    script = tmp_path / "api_worker.py"
    script.write_text(API_WORKER)
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"w{world}"
        out.mkdir()
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(D2R_REPO=REPO, D2R_OUT=str(out))
        cmd = [sys.executable, str(script)] if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
             "--master-port", "29713", str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[world] = out
    one = np.load(outs[1] / "api_scores_0.npy")
    assert (one != 0).sum() == 65
    for rank in (0, 1):
        np.testing.assert_array_equal(np.load(outs[2] / f"api_scores_{rank}.npy"), one)
        np.testing.assert_array_equal(np.load(outs[2] / f"api_best_{rank}.npy"), np.load(outs[1] / "api_best_0.npy"))
    assert ast.literal_eval((outs[1] / "api_calls_0.txt").read_text()) == [(65, 0, True)]
    assert ast.literal_eval((outs[2] / "api_calls_0.txt").read_text()) == [(33, 0, False)]
    assert ast.literal_eval((outs[2] / "api_calls_1.txt").read_text()) == [(32, 33, False)]
    files = sorted(os.listdir(outs[2] / "cb_render"))
    assert files == [f"cb_rgb_{i:04d}.png" for i in range(65)]
    assert [(outs[2] / "cb_render" / f).read_text() for f in files] == ["0"] * 33 + ["1"] * 32
    assert (outs[1] / "best_render.png").exists() and (outs[2] / "best_render.png").exists()
