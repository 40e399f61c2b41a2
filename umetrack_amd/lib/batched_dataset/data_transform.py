"""torch_data pre-processing (SURVEY.md section 8, rows A1' and F3).

Same names and argument meaning as lib/batched_dataset/data_transform.py:24-397: one raw
sequence (``mono`` images + ``labels``) becomes ``(ModelInput, ModelTarget)``.  The crop cameras
are host geometry (numpy, as in the reference); the image resampling - the only part with
per-pixel work - runs on the GPU in ``umt_resample_pinhole`` (csrc/warp.hip), which evaluates
the same fp32 homography, the same keep rule ``0 <= x < w-1 and 0 <= y < h-1`` and the same
four-tap float blend as data_transform.py:78-144.  The sequence's images therefore go to HBM
once as they come off the disk and the crops stay there.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from ... import _native as nat
from ..common import camera, crop
from ..common.hand import HandModel, mirrored_hand_model
from ..data_utils import bundles
from .sample import RawSample, parse_raw_buffers

scalar_type = np.float32


@dataclass
class PoseData:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    left_hand_model: HandModel


@dataclass
class ModelInput:
    orig_pose_data: PoseData
    s_solved_pose_data: PoseData
    left_images: torch.Tensor
    intrinsics: torch.Tensor
    extrinsics_xf: torch.Tensor
    hand_idx: torch.Tensor


@dataclass
class PerBranchOutput:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    skel_scales: Optional[torch.Tensor] = None
    pinch_prediction: Optional[torch.Tensor] = None


@dataclass
class ModelTarget:
    gt_skel_targets: PerBranchOutput
    preds_targets: PerBranchOutput
    intrinsics: Optional[torch.Tensor] = None
    extrinsics_xf: Optional[torch.Tensor] = None


def resample_images_batched(images_orig: torch.Tensor, resample_xfs: torch.Tensor,
                            crop_size: Tuple[int, int] = (96, 96)) -> torch.Tensor:
    """images_orig [n,h,w] float32, resample_xfs [n,4,4] float32 -> [n,H,W] float32 (not /255).
    Argument meaning of _resample_images_batched (data_transform.py:78-144); returns the new
    images instead of filling an array in place."""
    dev = nat.require_gpu(images_orig.device if images_orig.is_cuda else None)
    img = nat.dev_tensor(images_orig, torch.float32, dev)
    xf = nat.dev_tensor(resample_xfs, torch.float32, dev)
    n, h, w = img.shape
    assert tuple(xf.shape) == (n, 4, 4)
    out = torch.empty(n, crop_size[0], crop_size[1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().umt_resample_pinhole(nat.ptr(img), nat.ptr(xf), n, h, w, crop_size[0], crop_size[1],
                                                 nat.ptr(out), nat.stream_ptr(dev)), "umt_resample_pinhole")
    return out


def compute_resample_matrix(camera_orig: camera.PinholePlaneCameraModel,
                            camera_new: camera.PinholePlaneCameraModel) -> np.ndarray:
    """Pixel (u,v,1,1) of camera_new -> homogeneous pixel of camera_orig, rounded to fp32
    (data_transform.py:57-75): K_orig . world_to_eye_orig . eye_to_world_new . K_new^-1."""
    k_new_inv = np.eye(4)
    k_new_inv[:3, :3] = np.linalg.inv(camera_new.uv_to_window_matrix())
    k_orig = np.eye(4)
    k_orig[:3, :3] = camera_orig.uv_to_window_matrix()
    xf = k_orig @ np.linalg.inv(camera_orig.camera_to_world_xf) @ camera_new.camera_to_world_xf @ k_new_inv
    return xf.astype(np.float32)


def gen_crop_matrices(orig_extrinsics: np.ndarray, orig_intrinsics: np.ndarray, crop_points: np.ndarray,
                      mirror_image: bool, crop_size: Tuple[int, int], focal_multiplier: float = 0.95
                      ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Per view of one frame: crop-camera extrinsics (world->eye), intrinsics and the resample
    matrix (data_transform.py:147-213).  Raises ValueError for degenerate crops like the reference."""
    n_views = orig_extrinsics.shape[0]
    ext = np.empty((n_views, 4, 4), dtype=scalar_type)
    intr = np.empty((n_views, 3, 3), dtype=scalar_type)
    res = np.empty((n_views, 4, 4), dtype=scalar_type)
    for v in range(n_views):
        k = orig_intrinsics[v]
        cam_orig = camera.PinholePlaneCameraModel(width=crop_size[0], height=crop_size[1], f=(k[0, 0], k[1, 1]),
                                                  c=(k[0, 2], k[1, 2]), distort_coeffs=[],
                                                  camera_to_world_xf=np.linalg.inv(orig_extrinsics[v]))
        cam_new = crop.gen_crop_parameters_from_points(cam_orig, crop_points, crop_size, mirror_image,
                                                       camera_angle=0, focal_multiplier=focal_multiplier)
        ext[v] = np.linalg.inv(cam_new.camera_to_world_xf)
        intr[v] = cam_new.uv_to_window_matrix()
        res[v] = compute_resample_matrix(cam_orig, cam_new)
    return ext, intr, res


def perspective_crop_images(orig_images, orig_extrinsics: np.ndarray, orig_intrinsics: np.ndarray,
                            crop_points: np.ndarray, hand_idx: int, crop_size: Tuple[int, int],
                            device=None) -> List[Any]:
    """(T, views, H, W) images -> [crops (T, views, *crop_size) in 0..1 ON THE GPU, extrinsics_xf,
    intrinsics] (data_transform.py:216-281)."""
    n_frames, n_views = orig_images.shape[:2]
    ext = np.empty((n_frames, n_views, 4, 4), dtype=scalar_type)
    intr = np.empty((n_frames, n_views, 3, 3), dtype=scalar_type)
    res = np.empty((n_frames, n_views, 4, 4), dtype=scalar_type)
    for t in range(n_frames):
        ext[t], intr[t], res[t] = gen_crop_matrices(orig_extrinsics[t], orig_intrinsics[t], crop_points[t],
                                                    hand_idx == 1, crop_size)
    dev = nat.require_gpu(device)
    # np.array: the dataset hands out read-only views of a memory-mapped file
    imgs = torch.from_numpy(np.array(orig_images)) if not isinstance(orig_images, torch.Tensor) else orig_images
    imgs = imgs.to(dev).to(torch.float32).reshape(n_frames * n_views, *imgs.shape[2:])
    crops = resample_images_batched(imgs, torch.from_numpy(res.reshape(-1, 4, 4)), crop_size)
    crops = (crops / 255).reshape(n_frames, n_views, *crop_size)          # data_transform.py:279
    return [crops, ext, intr]


def prepare_inputs_targets(sample: RawSample, crop_size: Tuple[int, int], device=None
                           ) -> Tuple[ModelInput, ModelTarget]:
    """data_transform.py:284-385.  ``left_images`` lives on the GPU; the small label tensors stay
    on the host like the reference's (the caller moves them with ``bundles.to_device``)."""
    def to_th(a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.asarray(a)).float()

    sample.scaled(0.001)                                                   # mm -> m
    seq_length = sample.images.shape[0]

    def over_sequence(t: torch.Tensor) -> torch.Tensor:
        return t.unsqueeze(0).expand(seq_length, *t.shape)

    is_right = to_th(sample.hand) == 1
    generic = bundles.map_fields(over_sequence, sample.generic_hand_model, only_type=torch.Tensor)
    gt_model = bundles.map_fields(over_sequence, sample.hand_model, only_type=torch.Tensor)
    solved = PoseData(wrist_xfs=to_th(sample.solved_wrist_xfs), joint_angles=to_th(sample.solved_joint_angles),
                      left_hand_model=mirrored_hand_model(generic, is_right))
    orig = PoseData(wrist_xfs=to_th(sample.wrist), joint_angles=to_th(sample.joint_angles),
                    left_hand_model=mirrored_hand_model(gt_model, is_right))
    left_images, extrinsics_xf, intrinsics = perspective_crop_images(
        sample.images, sample.extrinsics, sample.intrinsics, sample.enclosing_points, int(sample.hand[0]),
        crop_size, device)
    model_input = ModelInput(orig_pose_data=orig, s_solved_pose_data=solved, left_images=left_images,
                             intrinsics=to_th(intrinsics), extrinsics_xf=to_th(extrinsics_xf),
                             hand_idx=to_th(sample.hand))
    pinch = to_th(sample.pinch)
    target = ModelTarget(
        gt_skel_targets=PerBranchOutput(orig.joint_angles, orig.wrist_xfs, orig.left_hand_model.hand_scale, pinch),
        preds_targets=PerBranchOutput(solved.joint_angles, solved.wrist_xfs, solved.left_hand_model.hand_scale, pinch),
        intrinsics=to_th(intrinsics), extrinsics_xf=to_th(extrinsics_xf))
    return model_input, target


def preprocess(data: Dict[str, Any], crop_size: Tuple[int, int], device=None) -> Tuple[ModelInput, ModelTarget]:
    """One dataset item ``{"mono": ..., "labels": ...}`` -> model input and ground truth
    (data_transform.py:388-397)."""
    return prepare_inputs_targets(parse_raw_buffers(**data), crop_size, device)
