"""bench.py's workload table and shard plan (host logic, no GPU): every BASELINE.json config is a named workload
whose pose grid is the one SURVEY.md section 8(d) lists, and the shards of every GPU count tile the grid."""
import json
import os

import numpy as np
import pytest

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_table_matches_baseline_json():
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert sorted(bench.BASELINE_CONFIGS) == list(range(len(base["configs"])))
    want = {0: (32, "shopping", (160, 90), "vit_b16"), 1: (4096, "shopping", (640, 360), "vit_b16"),
            2: (16384, "pool_triangle", (640, 360), "vit_b16"), 3: (131072, "shopping", (640, 360), "vit_b16"),
            4: (262144, "shelf", (640, 360), "vit_l14")}
    for k, c in bench.BASELINE_CONFIGS.items():
        n, scene, wh, clip = want[k]
        assert int(np.prod(c["sample_res"])) == n and c["scene"] == scene and (c["width"], c["height"]) == wh and c["clip"] == clip
        text = base["configs"][k].replace(" ", "").replace(" ", "")
        assert str(n) in text.replace(" ", "") or f"{n:,}".replace(",", "") in text
    assert bench.BASELINE_CONFIGS[4]["sample_res"] == [16, 16, 16, 4, 4, 4]       # 6-DoF, reference obj_pose_opt.py:22-29
    assert bench.BASELINE_CONFIGS[3]["sample_res"] == [128, 128, 8, 1, 1, 1]


@pytest.mark.parametrize("k", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shards_tile_the_grid(k, world):
    c = bench.BASELINE_CONFIGS[k]
    res = list(c["sample_res"])
    if c["scaling"] == "weak":
        res[2] = world
    N = int(np.prod(res))
    partition = max(world, c.get("partition", 1)) if c["scaling"] == "strong" else world
    plan = bench.shard_plan(res, world, partition)
    sizes = [len(s) for s in plan["shards"]]
    assert len(sizes) == world and max(sizes) - min(sizes) <= 1
    assert len(np.unique(plan["run_idx"])) == plan["n_run"] == sum(sizes)
    if partition == world:
        assert plan["n_run"] == N and np.array_equal(np.sort(plan["run_idx"]), np.arange(N))
    else:                                                   # config 4 below 8 GPUs: shards 0..world-1 of the 8-way partition
        assert plan["n_run"] == N * world // 8
    if c["scaling"] == "weak":                              # a rank owns whole (x, y) sheets: its poses share one z
        for r, s in enumerate(plan["shards"]):
            assert (s % world == r).all()
    if k == 4:                                              # 6-DoF: contiguous pose-order blocks, all 64 orientations of a position together
        s = plan["shards"][0]
        assert (np.diff(s) == 1).all() and len(s) % 64 == 0


def test_ragged_partitions_gather_in_rank_order():
    from dream2real_amd.dist import shard_range
    plan = bench.shard_plan([5, 3, 1, 1, 1, 1], 2, 4)       # 15 poses, 4-way partition, 2 ranks run: 4 + 4 of them
    assert [len(s) for s in plan["shards"]] == [4, 4] and plan["n_run"] == 8
    # the gather object splits n_run over the ranks exactly as the shards are sized
    assert [shard_range(8, r, 2) for r in range(2)] == [(0, 4), (4, 8)]
    plan = bench.shard_plan([5, 2, 1, 1, 1, 1], 3, 3)       # 10 over 3: 4, 3, 3
    assert [len(s) for s in plan["shards"]] == [4, 3, 3]


def _plan(*flags):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env["D2R_WATCHDOG_SCALE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--plan", *flags], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"plan"')]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


def test_eight_rank_plan_on_a_cpu_box():
    """`bench.py --gpus 8 --plan`: the 8-GPU scaling run worked out without a GPU (self-launch under torch.distributed.run,
    gloo): rank r takes device r (LOCAL_RANK), the shards are the eight (x, y) sheets of the weak-scaling grid, every rank holds
    the 1-GPU run's workspaces and nothing more, each rank's OpenMP / PNG workers get an eighth of the CPU quota, and the only
    data-path collective is the one all-gather of logits."""
    one, eight = _plan(), _plan("--gpus", "8")
    assert one["n_gpus"] == 1 and eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["sample_res"] == [64, 64, 8, 1, 1, 1]
    assert eight["poses_per_step"] == eight["poses_total"] == 8 * 4096
    ranks = eight["plan"]
    assert [e["rank"] for e in ranks] == list(range(8)) and [e["device"] for e in ranks] == [e["local_rank"] for e in ranks] == list(range(8))
    assert all(e["poses"] == 4096 and e["passes_per_step"] == 1 and e["chunk"] == 4096 for e in ranks)
    w1 = one["plan"][0]["workspace_bytes"]
    assert all(e["workspace_bytes"] == w1 for e in ranks)                       # per-rank footprint does not depend on the rank count
    assert w1["ray_queue"] == w1["ray_queue_sorted"] == 4096 * 640 * 360 * 8 and 30e9 < w1["total"] < 40e9 < 288e9
    quota = ranks[0]["cpu_threads"]["quota"]
    assert all(e["cpu_threads"]["local_world"] == 8 and e["cpu_threads"]["omp"] == e["cpu_threads"]["share_of_quota"] == max(1, quota // 8) for e in ranks)
    assert "one all-gather of 32768 x 2 fp32 logits = 262144 bytes" in ranks[0]["collective"]
    assert one["plan"][0]["cpu_threads"]["omp"] == quota


def test_strong_scaling_plan_splits_config3_and_config4():
    p3 = _plan("--gpus", "4", "--config", "3")
    assert p3["scaling"] == "strong" and p3["poses_total"] == 131072 and [e["poses"] for e in p3["plan"]] == [32768] * 4
    assert all(e["passes_per_step"] == 8 for e in p3["plan"])
    p4 = _plan("--gpus", "2", "--config", "4")                                   # two GPUs of the 8-way partition: shards 0 and 1
    assert p4["partition"] == 8 and p4["poses_per_step"] == 2 * 32768 and p4["clip"] == "vit_l14"
    assert [(e["first_pose"], e["last_pose"]) for e in p4["plan"]] == [(0, 32767), (32768, 65535)]
