"""Development diagnostic (GPU box): which pixels of a cone-stepped model's render have a hit on one side only (GPU vs
oracle), and what the two sides see there.  Uses the oracle: lives under tests/."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dream2real_amd import engine
from tests.scenes import make_scene
from tests.parity_utils import OraclePipeline
from oracle import host_ref

for kind in sys.argv[1:] or ["shelf", "room"]:
    scene = make_scene(kind)
    ctx = engine.Context(0)
    bg = engine.Testbed(ctx, scene.bg)
    W, H = 128, 72
    pipe = OraclePipeline(scene, W, H)
    cam = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    for opt in ((), (("bricks", 0),), (("raygen_rect", 0),)):
        for k, v in opt:
            ctx.set_option(k, v)
        rgba, depth = bg.render_batch(cam[None, :3], W, H)
        orgba, odepth = pipe.background()
        bad = np.argwhere((depth[0] > 0) != (odepth > 0))
        print(kind, opt, "samples gpu", bg.last_samples, "mismatching pixels", len(bad))
        for y, x in bad[:12]:
            print(f"  ({x},{y}) gpu depth {depth[0][y, x]:.6f} alpha {rgba[0][y, x, 3]:.6f} | oracle depth {odepth[y, x]:.6f} alpha {orgba[y, x, 3]:.6f}")
        for k, v in opt:
            ctx.set_option(k, 1)
    d = np.abs(depth[0] - odepth)
    ok = (depth[0] > 0) == (odepth > 0)
    print(kind, "max |ddepth| on agreeing pixels", d[ok].max(), "max |drgba|", np.abs(rgba[0] - orgba)[ok].max())
    bg.close(); ctx.close()

# composited candidate frames of the cone-stepped scenes: how far from the oracle, in LSB
from dream2real_amd import obj_pose_opt
from tests.scenes import make_task
for kind, res, wh in (("shelf", [3, 2, 1, 1, 1, 1], (128, 72)), ("room", [2, 2, 2, 1, 1, 1], (128, 72)), ("shelf", [2, 2, 2, 3, 2, 2], (128, 72))):
    scene = make_scene(kind)
    ctx = engine.Context(0)
    fg = engine.Testbed(ctx, scene.fg)
    fg.background_color = list(scene.fg_background)
    W, H = wh
    pipe = OraclePipeline(scene, W, H)
    poses = host_ref.sample_poses_grid(scene.scene_centre, res, scene.scene_type).reshape(-1, 4, 4)
    obg = pipe.background()
    view = fg.view(W, H)
    ctx.set_background(view, obg[0], obg[1])
    T1 = host_ref.converter(np.asarray(scene.obj_pose, np.float32)[None])[0]
    TC = host_ref.converter(np.asarray(scene.cam_poses, np.float32))[0]
    frames = fg.render_composite(view, T1, TC, host_ref.converter(poses.astype(np.float32)))
    want = pipe.frames(poses, bg=obg)
    diff = np.abs(frames.astype(int) - want.astype(int)).max(-1)
    print(kind, res, "frames", len(poses), "max LSB", diff.max(), "pixels > 1 LSB", int((diff > 1).sum()), "of", diff.size, "frac off by >= 1", float((diff > 0).mean()),
          "hist", np.bincount(diff.reshape(-1))[:8].tolist())
    fg.close(); ctx.close()
