"""Run on the GPU box: zlib strategies on REAL rendered frames (Sub-filtered scanlines), one thread: MB/s and bytes per frame."""
import os, sys, time, zlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_api_path import _setup
from dream2real_amd import combined_rendering
from dream2real_amd.accio2ngp import converter
from dream2real_amd.obj_pose_opt import sample_poses_grid
from dream2real_amd.virtual_cam_pose_sample import get_virtual_cam_poses
for (W, H) in ((640, 360), (336, 336)):
    scene, ctx, fg, bg, sc, task, text = _setup(W, H, "vit_tiny")
    poses = converter(sample_poses_grid(task, [8, 8, 1, 1, 1, 1], scene.scene_type).reshape(-1, 4, 4))
    rp = converter(get_virtual_cam_poses(task, [0]))
    rend = combined_rendering.renderer(tempfile.mkdtemp(), task, resolution=(W, H))
    frames = np.stack(rend.render(poses, rp, [0], save=False))
    f = frames.astype(np.int16)
    sub = f.copy(); sub[:, :, 1:] -= f[:, :, :-1]
    sub = sub.astype(np.uint8)
    raw = [np.concatenate([np.ones((H, 1), np.uint8), s.reshape(H, -1)], 1).tobytes() for s in sub]
    raw0 = [np.concatenate([np.zeros((H, 1), np.uint8), s.reshape(H, -1)], 1).tobytes() for s in frames]
    for name, data, lvl, strat in (("none+default L1", raw0, 1, zlib.Z_DEFAULT_STRATEGY), ("sub+RLE L1", raw, 1, zlib.Z_RLE), ("sub+HUFFMAN", raw, 1, zlib.Z_HUFFMAN_ONLY),
                                   ("sub+FILTERED L1", raw, 1, zlib.Z_FILTERED), ("sub+default L1", raw, 1, zlib.Z_DEFAULT_STRATEGY), ("sub+FIXED L1", raw, 1, zlib.Z_FIXED), ("stored", raw0, 0, zlib.Z_DEFAULT_STRATEGY)):
        t = time.time(); n = 0
        for d in data:
            c = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, strat)
            n += len(c.compress(d)) + len(c.flush())
        dt = time.time() - t
        print(f"{W}x{H} {name:18s} {len(data) * len(data[0]) / dt / 1e6:7.1f} MB/s  {n / len(data) / 1024:7.1f} KiB/frame", flush=True)
    sc.close(); fg.close(); bg.close(); ctx.close()
