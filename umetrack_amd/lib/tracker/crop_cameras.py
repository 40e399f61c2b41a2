"""Batched crop-camera generation on the GPU.

``gen_crop_cameras_batched`` does for a whole batch of hands what
HandTracker.gen_crop_cameras -> gen_crop_cameras_from_pose does per hand on the host
(lib/tracker/tracker.py:101-139, lib/tracker/perspective_crop.py:54-162): FK of the tracked,
neutral and open pose (one ``umt_fk_landmarks`` launch), visibility ranking of the rig cameras,
look-at crop cameras, and the K / E packing of ``_make_inputs`` (tracker.py:212-216).  Every
output stays on the device in the layouts ``umt_warp_crops`` and the model consume.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from ... import _native as nat
from ..common.hand import HandModel
from ..common.hand_skinning import skin_landmarks
from .perspective_crop import neutral_joint_angles


@dataclass
class CropCameraBatch:
    n_views: torch.Tensor      # [B] int32
    view_idx: torch.Tensor     # [B, max_views] int32, -1 = unused slot
    dst_cams: torch.Tensor     # [B, max_views, 20] float64 (fx, fy, cx, cy, cam_to_world)
    intrinsics: torch.Tensor   # [B, max_views, 3, 3] float32
    extrinsics: torch.Tensor   # [B, max_views, 4, 4] float32, world -> crop eye, translation in metres
    status: torch.Tensor       # [B, max_views] int32, 1 = the reference would raise ValueError
    crop_points: torch.Tensor  # [B, P, 3] float32 world mm


def crop_points_batched(hand_model: HandModel, joint_angles: torch.Tensor, wrist_xforms: torch.Tensor,
                        hand_idx: torch.Tensor, num_crop_points: int = 63) -> torch.Tensor:
    """[B,22], [B,4,4] (left-model convention, mm), [B] -> [B, num_crop_points, 3]: landmarks of the
    tracked pose, then the neutral pose, then the open hand (perspective_crop.py:89-115)."""
    assert num_crop_points in (21, 42, 63)
    dev = nat.require_gpu(joint_angles.device if joint_angles.is_cuda else None)
    ja = nat.dev_tensor(joint_angles, torch.float32, dev)
    B = ja.shape[0]
    poses = [ja]
    if num_crop_points > 21:
        poses.append(neutral_joint_angles(hand_model).to(dev, torch.float32).expand(B, 22))
    if num_crop_points > 42:
        poses.append(torch.zeros(B, 22, device=dev))
    n_pose = len(poses)
    allp = torch.stack(poses, 1).reshape(B * n_pose, 22)
    wr = nat.dev_tensor(wrist_xforms, torch.float32, dev).repeat_interleave(n_pose, 0)
    hidx = nat.dev_tensor(hand_idx, torch.int64, dev).repeat_interleave(n_pose, 0)
    return skin_landmarks(hand_model, allp, wr, hand_idx=hidx, mirror_right=True).reshape(B, num_crop_points, 3)


def gen_crop_cameras_batched(crop_points: torch.Tensor, cameras: torch.Tensor, camera_angles: torch.Tensor,
                             hand_idx: torch.Tensor, rig_idx: Optional[torch.Tensor] = None,
                             forced_views: Optional[torch.Tensor] = None, new_image_size: Tuple[int, int] = (96, 96),
                             max_view_num: int = 2, sort_camera_index: bool = True, focal_multiplier: float = 0.95,
                             mirror_right_hand: bool = True, min_required_vis_landmarks: int = 19) -> CropCameraBatch:
    """crop_points [B,P,3] float32 (world mm); cameras [V,30] or [rigs,V,30] float64 in
    ``warp.pack_fisheye`` format; camera_angles [V] or [rigs,V] degrees.  Defaults are the tracker's
    (tracker.py:116-129)."""
    dev = nat.require_gpu(crop_points.device if crop_points.is_cuda else None)
    pts = nat.dev_tensor(crop_points, torch.float32, dev)
    B, P = pts.shape[0], pts.shape[1]
    cams = nat.dev_tensor(cameras, torch.float64, dev)
    if cams.dim() == 2:
        cams = cams[None]
    ang = nat.dev_tensor(camera_angles, torch.float64, dev).reshape(cams.shape[0], -1)
    V = cams.shape[1]
    if cams.shape[2] != 30 or ang.shape[1] != V:
        raise ValueError(f"cameras must be [rigs,V,30] with one angle per camera, got {tuple(cams.shape)} / {tuple(ang.shape)}")
    hidx = nat.dev_tensor(hand_idx, torch.int64, dev)
    rig = None if rig_idx is None else nat.dev_tensor(rig_idx, torch.int32, dev)
    if rig is None and cams.shape[0] != 1:
        raise ValueError("several rigs need rig_idx")
    forced = None if forced_views is None else nat.dev_tensor(forced_views, torch.int32, dev).reshape(B, max_view_num)
    mv = max_view_num
    out = CropCameraBatch(torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, mv, dtype=torch.int32, device=dev),
                          torch.empty(B, mv, 20, dtype=torch.float64, device=dev),
                          torch.empty(B, mv, 3, 3, dtype=torch.float32, device=dev),
                          torch.empty(B, mv, 4, 4, dtype=torch.float32, device=dev),
                          torch.empty(B, mv, dtype=torch.int32, device=dev), pts)
    with torch.cuda.device(dev):
        nat.check(nat.lib().umt_crop_cameras(
            nat.ptr(pts), P, nat.ptr(cams), nat.ptr(ang), V, nat.ptr(rig), nat.ptr(hidx), nat.ptr(forced), B, mv,
            int(min_required_vis_landmarks), int(sort_camera_index), int(mirror_right_hand), float(focal_multiplier),
            int(new_image_size[0]), int(new_image_size[1]), nat.ptr(out.n_views), nat.ptr(out.view_idx),
            nat.ptr(out.dst_cams), nat.ptr(out.intrinsics), nat.ptr(out.extrinsics), nat.ptr(out.status),
            nat.stream_ptr(dev)), "umt_crop_cameras")
    return out
