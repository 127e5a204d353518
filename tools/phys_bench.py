"""Throughput of the GPU physics pre-filter on a shopping-sized grid (70 000 poses, configs/shopping_demo.json:28)
with hulls of a few dozen vertices.  Development tool; run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dream2real_amd import engine, physics_utils, obj_pose_opt
from synthetic_scenes import box, icosphere
import types, torch

ctx = engine.Context(0)
centre = np.array([0.5, 0.0, 0.035])
mov = icosphere(centre + [0.0, -0.05, 0.05], 0.045, 64, 1)
statics = [box([-0.2, -0.6, -0.06], [1.2, 0.6, 0.0]), icosphere(centre + [0.12, -0.10, 0.05], 0.05, 80, 2),
           icosphere(centre + [-0.15, 0.08, 0.04], 0.04, 80, 3), icosphere(centre + [0.05, 0.15, 0.045], 0.045, 80, 4)]
init = np.eye(4, dtype=np.float32)
init[:3, 3] = centre + [0.0, -0.05, 0.05]
res = [100, 100, 7, 1, 1, 1]
task = types.SimpleNamespace(scene_model=types.SimpleNamespace(scene_centre=torch.tensor(centre, dtype=torch.float32)))
poses = obj_pose_opt.sample_poses_grid(task, res, 3)
import json
sh = physics_utils.PhysicsShapes(ctx, mov, statics)
out = {"workload": "shopping-sized grid [100,100,7,1,1,1] = 70000 poses (configs/shopping_demo.json:28), movable hull of 64 vertices, 4 static hulls (8 + 3 x 80 vertices)",
       "includes": "host -> device pose upload (4.5 MB) and the mask read-back", "runs": []}
for margin in (0.0, physics_utils.PYBULLET_MESH_MARGIN):
    v = sh.check(poses, np.ones(len(poses), bool), res, init, float(centre[2]) - 0.1, margin=margin)
    t0 = time.perf_counter()
    for _ in range(5):
        v = sh.check(poses, np.ones(len(poses), bool), res, init, float(centre[2]) - 0.1, margin=margin)
    dt = (time.perf_counter() - t0) / 5
    out["runs"].append({"margin_m": margin, "ms": round(dt * 1e3, 3), "poses_per_s": round(len(poses) / dt), "valid_fraction": round(float(v.mean()), 4)})
print(json.dumps(out))
