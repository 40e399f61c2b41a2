"""Crop-camera selection per hand (row A0), names as in lib/tracker/perspective_crop.py:19-162.

Host geometry is numpy float64 like the reference; the landmark positions it needs
come from the GPU forward-kinematics kernel (lib.common.hand_skinning.skin_landmarks).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..common import camera
from ..common.crop import gen_crop_parameters_from_points
from ..common.hand import NUM_JOINTS_PER_HAND, RIGHT_HAND_INDEX, HandModel
from ..common.hand_skinning import skin_landmarks
from .tracking_result import SingleHandPose


def neutral_joint_angles(up: HandModel, lower_factor: float = 0.5) -> torch.Tensor:
    lim = up.joint_limits
    assert lim is not None
    return lim[..., 0] * lower_factor + lim[..., 1] * (1 - lower_factor)


def skin_landmarks_np(hand_model: HandModel, joint_angles: np.ndarray, wrist_transforms: np.ndarray) -> np.ndarray:
    out = skin_landmarks(hand_model, torch.from_numpy(np.asarray(joint_angles)).float(),
                         torch.from_numpy(np.asarray(wrist_transforms)).float())
    return out.cpu().numpy()


def landmarks_from_hand_pose(hand_model: HandModel, hand_pose: SingleHandPose, hand_idx: int) -> np.ndarray:
    """World-space landmarks of a (left-model) hand pose; right hands mirror the wrist x axis
    (perspective_crop.py:40-51).  A pose that HandTracker.track_frame returned already carries the landmarks the
    GPU computed for it in the frame's launch chain; they are returned when `hand_model` is the model they were
    skinned with (same object) and the pose has not been edited since."""
    cached = getattr(hand_pose, "landmarks_mm", None)
    if cached is not None and getattr(hand_pose, "landmarks_model", None) is hand_model \
            and getattr(hand_pose, "landmarks_hand_idx", -1) == hand_idx:
        return cached
    xf = np.array(hand_pose.wrist_xform, copy=True)
    if hand_idx == RIGHT_HAND_INDEX:
        xf[:, 0] *= -1
    return skin_landmarks_np(hand_model, hand_pose.joint_angles, xf)


def crop_points_and_landmarks(hand_model: HandModel, hand_pose: SingleHandPose, hand_idx: int,
                              num_crop_points: int) -> Tuple[np.ndarray, np.ndarray]:
    """One FK launch for the (up to) three poses that bound the crop: tracked, neutral, open
    (perspective_crop.py:89-115).  Returns (crop_points [num_crop_points,3], landmarks [21,3])."""
    assert num_crop_points in (21, 42, 63)
    poses = [np.asarray(hand_pose.joint_angles, np.float32)]
    if num_crop_points > 21:
        poses.append(neutral_joint_angles(hand_model).numpy().astype(np.float32))
    if num_crop_points > 42:
        poses.append(np.zeros(NUM_JOINTS_PER_HAND, np.float32))
    xf = np.array(hand_pose.wrist_xform, copy=True)
    if hand_idx == RIGHT_HAND_INDEX:
        xf[:, 0] *= -1
    lm = skin_landmarks_np(hand_model, np.stack(poses), np.broadcast_to(xf, (len(poses), 4, 4)).copy())
    return lm.reshape(-1, 3), lm[0]


def count_visible(cam: camera.CameraModel, landmarks_world: np.ndarray) -> int:
    eye = cam.world_to_eye(landmarks_world)
    win = cam.eye_to_window(eye)
    return int(((win[..., 0] >= 0) & (win[..., 0] <= cam.width - 1) & (win[..., 1] >= 0)
                & (win[..., 1] <= cam.height - 1) & (eye[..., 2] > 0)).sum())


def rank_hand_visibility_in_cameras(cameras: List[camera.CameraModel], hand_model: HandModel,
                                    hand_pose: SingleHandPose, hand_idx: int, min_required_vis_landmarks: int,
                                    landmarks_world: Optional[np.ndarray] = None) -> List[int]:
    """Cameras that see enough landmarks, most landmarks first (perspective_crop.py:54-86)."""
    if landmarks_world is None:
        landmarks_world = landmarks_from_hand_pose(hand_model, hand_pose, hand_idx)
    seen = [count_visible(c, landmarks_world) for c in cameras]
    keep = [i for i, n in enumerate(seen) if n >= min_required_vis_landmarks]
    return sorted(keep, key=lambda i: seen[i], reverse=True)     # stable, like list.sort


def gen_crop_cameras_from_pose(cameras: List[camera.CameraModel], camera_angles: List[float], hand_model: HandModel,
                               hand_pose: SingleHandPose, hand_idx: int, num_crop_points: int,
                               new_image_size: Tuple[int, int], max_view_num: Optional[int] = None,
                               sort_camera_index: bool = False, focal_multiplier: float = 0.95,
                               mirror_right_hand: bool = True,
                               min_required_vis_landmarks: int = 19) -> Dict[int, camera.PinholePlaneCameraModel]:
    """perspective_crop.py:118-162."""
    crop_points, landmarks = crop_points_and_landmarks(hand_model, hand_pose, hand_idx, num_crop_points)
    order = rank_hand_visibility_in_cameras(cameras, hand_model, hand_pose, hand_idx, min_required_vis_landmarks,
                                            landmarks_world=landmarks)
    if sort_camera_index:
        order = sorted(order)
    out: Dict[int, camera.PinholePlaneCameraModel] = {}
    for ci in order:
        # crop_points stay float32 like the reference's: the bbox centre is rounded in float32 there
        out[ci] = gen_crop_parameters_from_points(cameras[ci], crop_points, new_image_size,
                                                  mirror_img_x=(mirror_right_hand and hand_idx == 1),
                                                  camera_angle=camera_angles[ci], focal_multiplier=focal_multiplier)
        if len(out) == max_view_num:
            break
    return out
