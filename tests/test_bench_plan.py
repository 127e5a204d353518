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
