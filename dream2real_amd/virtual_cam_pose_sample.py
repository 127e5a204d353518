"""Render-camera selection (reference vision_3d/virtual_cam_pose_sample.py:5-8)."""
import numpy as np


def get_virtual_cam_poses(task_model, render_cam_pose_idx):
    """[L,4,4] optimised training-view camera poses picked by index."""
    poses = task_model.scene_model.opt_cam_poses
    return np.stack([np.asarray(poses[int(i)].cpu() if hasattr(poses[int(i)], "cpu") else poses[int(i)])
                     for i in render_cam_pose_idx], axis=0)
