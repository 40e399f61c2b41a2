"""Per-frame tracker with the reference's call surface (lib/tracker/tracker.py:26-291).

The image warp, the network and the forward kinematics run on the GPU; this class
only packs inputs and unpacks outputs.  ``HandTracker`` needs a model created by
``umetrack_amd.lib.models.model_loader`` and an MI355X.
"""
import logging
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from ..common import camera
from ..common.hand import NUM_HANDS, HandModel, scaled_hand_model
from ..models.regressor import RegressorOutput
from ..models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
from . import warp as gpu_warp
from .perspective_crop import gen_crop_cameras_from_pose
from .tracking_result import SingleHandPose, TrackingResult

logger = logging.getLogger(__name__)

MM_TO_M = 0.001
M_TO_MM = 1000.0
MIN_OBSERVED_LANDMARKS = 21
CONFIDENCE_THRESHOLD = 0.5
MAX_VIEW_NUM = 2
INTER_LINEAR = 1          # value of cv2.INTER_LINEAR, the only interpolation the reference uses


@dataclass
class ViewData:
    image: np.ndarray
    camera: camera.CameraModel
    camera_angle: float


@dataclass
class InputFrame:
    views: List[ViewData]


@dataclass
class HandTrackerOpts:
    num_crop_points: int = 63
    enable_memory: bool = True
    use_stored_pose_for_crop: bool = True
    hand_ratio_in_crop: float = 0.95
    min_required_vis_landmarks: int = 19
    warp_mode: int = gpu_warp.MODE_CV2     # OpenCV-compatible fixed-point sampling (see csrc/warp.hip)


def _pack_views(views, pairs, device):
    """Upload the source views used by `pairs` [(src view index, crop camera)] once and build the
    per-crop descriptors for umt_warp_crops."""
    used = sorted({vi for vi, _ in pairs})
    chunks, start, pos = [], {}, 0
    for vi in used:
        img = np.ascontiguousarray(views[vi].image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError(f"view {vi}: expected a 2-D uint8 image, got {img.dtype} {img.shape}")
        start[vi] = (pos, img.shape[1])
        chunks.append(img.reshape(-1))
        pos += img.size
    flat = torch.from_numpy(np.concatenate(chunks)).to(device)
    offs = [start[vi][0] for vi, _ in pairs]
    pitch = [start[vi][1] for vi, _ in pairs]
    src, dst = [], []
    for vi, crop_cam in pairs:
        cam = views[vi].camera
        if not isinstance(cam, camera.Fisheye62CameraModel):
            raise TypeError("the GPU warp supports Fisheye62 source cameras (the raw_data rigs)")
        src.append(gpu_warp.pack_fisheye(cam.width, cam.height, cam.f, cam.c, list(cam.distort), cam.camera_to_world_xf))
        dst.append(gpu_warp.pack_pinhole(crop_cam.f, crop_cam.c, crop_cam.camera_to_world_xf))
    return flat, offs, pitch, np.stack(src), np.stack(dst)


def _warp_image(src_camera: camera.CameraModel, dst_camera: camera.CameraModel, src_image: np.ndarray,
                interpolation: int = INTER_LINEAR, depth_check: bool = True) -> np.ndarray:
    """Single-image form of the warp (tracker.py:54-80): uint8 in, uint8 out, on the GPU."""
    if interpolation != INTER_LINEAR:
        raise NotImplementedError("only cv2.INTER_LINEAR (the reference's own call, and the only sampling whose "
                                  "fixed-point arithmetic is restated here) is provided")
    view = ViewData(src_image, src_camera, 0.0)
    flat, offs, pitch, src, dst = _pack_views([view], [(0, dst_camera)], "cuda")
    mode = gpu_warp.MODE_CV2 if depth_check else gpu_warp.MODE_CV2 | gpu_warp.NO_DEPTH_CHECK
    out = gpu_warp.warp_crops(flat, offs, pitch, src, dst, (dst_camera.height, dst_camera.width), mode)
    return torch.round(out[0] * 255.0).to(torch.uint8).cpu().numpy().astype(src_image.dtype)


class HandTracker:
    def __init__(self, model, opts: HandTrackerOpts) -> None:
        self._device = "cuda"
        self._model = model
        self._model.to(self._device)          # raises without an MI355X: there is no CPU path
        self._input_size = np.array(self._model.getInputImageSizes())
        self._num_crop_points = opts.num_crop_points
        self._enable_memory = opts.enable_memory
        self._hand_ratio_in_crop = opts.hand_ratio_in_crop
        self._min_required_vis_landmarks = opts.min_required_vis_landmarks
        self._warp_mode = opts.warp_mode
        self._valid_tracking_history = np.zeros(2, dtype=bool)

    def reset_history(self, clear_model_memory: bool = False) -> None:
        """tracker.py:98-99 clears the tracking history only (the next frame runs with use_memory=False, which
        neutralises the module memory).  ``clear_model_memory`` additionally drops the model's temporal rows."""
        self._valid_tracking_history[:] = False
        if clear_model_memory:
            self._model.reset_memory()

    def gen_crop_cameras(self, cameras: List[camera.CameraModel], camera_angles: List[float], hand_model: HandModel,
                         gt_tracking: Dict[int, SingleHandPose],
                         min_num_crops: int) -> Dict[int, Dict[int, camera.PinholePlaneCameraModel]]:
        """tracker.py:101-139."""
        out: Dict[int, Dict[int, camera.PinholePlaneCameraModel]] = {}
        for hand_idx, pose in (gt_tracking or {}).items():
            if pose.hand_confidence < CONFIDENCE_THRESHOLD:
                continue
            cams = gen_crop_cameras_from_pose(cameras, camera_angles, hand_model, pose, hand_idx, self._num_crop_points,
                                              self._input_size, max_view_num=MAX_VIEW_NUM, sort_camera_index=True,
                                              focal_multiplier=self._hand_ratio_in_crop, mirror_right_hand=True,
                                              min_required_vis_landmarks=self._min_required_vis_landmarks)
            if cams and len(cams) >= min_num_crops:
                out[hand_idx] = cams
        return out

    def track_frame(self, sample: InputFrame, hand_model: HandModel,
                    crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]) -> TrackingResult:
        if not crop_cameras:
            self.reset_history()
            return TrackingResult()
        frame_data, frame_desc, skeleton = self._make_inputs(sample, hand_model, crop_cameras)
        out = self._model.regress_pose_use_skeleton(frame_data, frame_desc, skeleton)
        return self._gen_tracking_result(out, frame_desc.hand_idx.cpu().numpy(), crop_cameras)

    def track_frame_and_calibrate_scale(self, sample: InputFrame,
                                        crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]) -> TrackingResult:
        if not crop_cameras:
            self.reset_history()
            return TrackingResult()
        frame_data, frame_desc, _ = self._make_inputs(sample, None, crop_cameras)
        out = self._model.regress_pose_pred_skel_scale(frame_data, frame_desc)
        return self._gen_tracking_result(out, frame_desc.hand_idx.cpu().numpy(), crop_cameras)

    def _make_inputs(self, sample: InputFrame, hand_model_mm: Optional[HandModel],
                     crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]):
        """tracker.py:194-247; all crops of the frame are warped in one kernel launch."""
        pairs, ranges, hands, K, E = [], [], [], [], []
        for hand_idx, per_hand in crop_cameras.items():
            first = len(pairs)
            for cam_idx, crop_cam in per_hand.items():
                pairs.append((cam_idx, crop_cam))
                K.append(crop_cam.uv_to_window_matrix())
                w2e = np.linalg.inv(crop_cam.camera_to_world_xf)
                w2e[:3, 3] *= MM_TO_M
                E.append(w2e)
            if len(pairs) > first:
                hands.append(hand_idx)
                ranges.append((first, len(pairs)))
        hands = np.asarray(hands)
        flat, offs, pitch, src, dst = _pack_views(sample.views, pairs, self._device)
        crops = gpu_warp.warp_crops(flat, offs, pitch, src, dst, tuple(int(v) for v in self._input_size),
                                    self._warp_mode)
        dev = self._device
        frame_data = InputFrameData(left_images=crops,
                                    intrinsics=torch.from_numpy(np.stack(K)).float().to(dev),
                                    extrinsics_xf=torch.from_numpy(np.stack(E)).float().to(dev))
        use_mem = self._valid_tracking_history[hands]      # enable_memory is stored but unused, as in the reference
        frame_desc = InputFrameDesc(sample_range=torch.from_numpy(np.asarray(ranges)).long().to(dev),
                                    memory_idx=torch.from_numpy(hands).long().to(dev),
                                    use_memory=torch.from_numpy(use_mem).bool().to(dev),
                                    hand_idx=torch.from_numpy(hands).long().to(dev))
        skeleton = None
        if hand_model_mm is not None:
            hm = scaled_hand_model(hand_model_mm, MM_TO_M)
            skeleton = InputSkeletonData(joint_rotation_axes=hm.joint_rotation_axes.float().to(dev),
                                         joint_rest_positions=hm.joint_rest_positions.float().to(dev))
        return frame_data, frame_desc, skeleton

    def _gen_tracking_result(self, out: RegressorOutput, hand_indices: np.ndarray,
                             crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]) -> TrackingResult:
        """tracker.py:249-291: metres -> millimetres, per-hand dicts, tracking history update."""
        angles = out.joint_angles.cpu().numpy()
        wrists = out.wrist_xfs.cpu().numpy()
        check = getattr(self._model, "check_status", None)
        if check is not None:               # the reference raises inside the forward pass; here after the copy-out
            check()
        wrists[..., :3, 3] *= M_TO_MM
        scales = None if out.skel_scales is None else out.skel_scales.cpu().numpy()
        poses, views, pred_scales = {}, {}, {}
        for row, hand_idx in enumerate(hand_indices):
            poses[hand_idx] = SingleHandPose(joint_angles=angles[row], wrist_xform=wrists[row], hand_confidence=1.0)
            views[hand_idx] = len(crop_cameras[hand_idx])
            if scales is not None:
                pred_scales[hand_idx] = scales[row]
        for hand_idx in range(NUM_HANDS):
            self._valid_tracking_history[hand_idx] = hand_idx in poses
        return TrackingResult(hand_poses=poses, num_views=views, predicted_scales=pred_scales)
