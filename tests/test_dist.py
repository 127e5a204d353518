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
