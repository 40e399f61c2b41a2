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
from ... import _native as nat
from ..common.hand import NUM_HANDS, HandModel, hand_model_to, scaled_hand_model
from ..common.hand_skinning import skin_landmarks
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


def _check_view(vi, img):
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError(f"view {vi}: expected a 2-D uint8 image, got {img.dtype} {img.shape}")


def _common_parent(images):
    """If the views are column blocks of the same rows of one array (the reference slices a decoded H x 4W frame into its
    four views, video_pose_data.py:123-125), the 2-D array (H, span) with their common row pitch that covers exactly
    those columns - contiguous when the views tile whole rows - else None.  Only memory inside the views' common base
    array is ever addressed."""
    base = images[0].base
    if base is None or not isinstance(base, np.ndarray):
        return None
    h, pitch = images[0].shape[0], images[0].strides[0]
    addr = []
    for img in images:
        if img.base is not base or img.shape[0] != h or img.strides != (pitch, 1):
            return None
        addr.append(img.__array_interface__["data"][0])
    a0 = min(addr)
    lo = images[addr.index(a0)]
    span = max(a + img.shape[1] for a, img in zip(addr, images)) - a0
    if pitch < span or span * 4 > 5 * sum(img.shape[1] for img in images):      # not one row block, or mostly unused columns
        return None
    b_lo, b_hi = np.lib.array_utils.byte_bounds(base) if hasattr(np.lib, "array_utils") else np.byte_bounds(base)
    if a0 < b_lo or a0 + (h - 1) * pitch + span > b_hi:
        return None
    return np.lib.stride_tricks.as_strided(lo, shape=(h, span), strides=(pitch, 1), writeable=False)


def _pack_views(views, pairs, device):
    """Upload the source views used by `pairs` [(src view index, crop camera)] once and build the
    per-crop descriptors for umt_warp_crops (the single-image form; HandTracker stages whole frames through
    _FrameIO instead)."""
    used = sorted({vi for vi, _ in pairs})
    chunks, start, pos = [], {}, 0
    for vi in used:
        img = np.ascontiguousarray(views[vi].image)
        _check_view(vi, img)
        start[vi] = (pos, img.shape[1])
        chunks.append(img.reshape(-1))
        pos += img.size
    flat = torch.from_numpy(np.concatenate(chunks)).to(device)
    offs = [start[vi][0] for vi, _ in pairs]
    pitch = [start[vi][1] for vi, _ in pairs]
    src, dst = [], []
    for vi, crop_cam in pairs:
        src.append(_pack_src(views[vi].camera))
        dst.append(gpu_warp.pack_pinhole(crop_cam.f, crop_cam.c, crop_cam.camera_to_world_xf))
    return flat, offs, pitch, np.stack(src), np.stack(dst)


def _pack_src(cam) -> np.ndarray:
    if not isinstance(cam, camera.Fisheye62CameraModel):
        raise TypeError("the GPU warp supports Fisheye62 source cameras (the raw_data rigs)")
    return gpu_warp.pack_fisheye(cam.width, cam.height, cam.f, cam.c, list(cam.distort), cam.camera_to_world_xf)


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


class TrackedHandPose(SingleHandPose):
    """A SingleHandPose (same three fields, same tuple) that also carries the landmarks the tracker computed on the
    GPU in the same launch chain as the pose: `landmarks_from_hand_pose` returns them when it is asked about the
    hand model they were skinned with, instead of a second round trip to the device per hand."""
    landmarks_mm: Optional[np.ndarray] = None
    landmarks_model: Optional[HandModel] = None
    landmarks_hand_idx: int = -1


def _align(v: int, a: int = 256) -> int:
    return (v + a - 1) // a * a


class _FrameIO:
    """Per-tracker staging: ONE pinned host block and its device twin.  Everything a frame sends to the GPU (source
    pixels, camera rows, K, E, index tensors) is written into the pinned block and goes up in two asynchronous copies
    (the pixels as soon as they are staged, the small rows behind them);
    everything the frame returns (angles, wrist transforms, sigmas, scales, landmarks, the model's status word)
    comes back in one copy.  The reference does the same traffic as ~10 separate pageable copies
    (tracker.py:239-247, 156-161), each of which blocks the host."""

    _IN = (("src", np.float64, gpu_warp.SRC_CAM_WORDS, "n"), ("dst", np.float64, gpu_warp.DST_CAM_WORDS, "n"),
           ("off", np.int64, 1, "n"), ("sr", np.int64, 2, "bs"), ("midx", np.int64, 1, "bs"), ("hidx", np.int64, 1, "bs"),
           ("K", np.float32, 9, "n"), ("E", np.float32, 16, "n"), ("pitch", np.int32, 1, "n"), ("use", np.uint8, 1, "bs"))
    _OUT = (("status", np.int32, 4, "one"), ("angles", np.float32, 22, "bs"), ("wrist", np.float32, 16, "bs"),
            ("sigmas", np.float32, 21, "bs"), ("scales", np.float32, 1, "bs"), ("lm", np.float32, 63, "bs"))
    _TORCH = {np.float64: torch.float64, np.int64: torch.int64, np.float32: torch.float32, np.int32: torch.int32,
              np.uint8: torch.uint8}

    def __init__(self, device):
        self.device = torch.device(device)
        self.cap = (0, 0, 0)
        self.pin = self.dev = None
        self.layout = {}

    def plan(self, n: int, bs: int, pixel_bytes: int):
        """Section offsets for n crops, bs hands and `pixel_bytes` of source pixels: pixels, camera rows and index
        tensors, then the status words (sent up as zeros, brought back with the outputs), then the outputs."""
        key = (n, bs, max(_align(pixel_bytes), self.cap[2]))     # the pixel span only grows: no re-plan when fewer views are used
        if key == self.cap and self.pin is not None:
            return self.layout
        count = {"n": n, "bs": bs, "one": 1}
        lay = {"pixels": (0, key[2], np.uint8, (key[2],))}
        pos = params_begin = _align(key[2])
        for name, dt, words, per in self._IN:
            nb = np.dtype(dt).itemsize * words * count[per]
            lay[name] = (pos, nb, dt, (count[per], words))
            pos = _align(pos + nb)
        lay["out_begin"] = pos
        for name, dt, words, per in self._OUT:
            nb = np.dtype(dt).itemsize * words * count[per]
            lay[name] = (pos, nb, dt, (count[per], words))
            if name == "status":
                lay["in_end"] = pos + nb
            pos = _align(pos + nb)
        lay["out_end"] = pos
        if self.pin is None or self.pin.numel() < pos:
            self.pin = torch.zeros(pos + (pos >> 2), dtype=torch.uint8).pin_memory()
            self.dev = torch.empty(self.pin.numel(), dtype=torch.uint8, device=self.device)
            self._host = self.pin.numpy()
        self.cap, self.layout = key, lay
        # typed views of every section, host and device side, made once per layout (a frame touches ~20 of them)
        self._hv, self._dv = {}, {}
        for name, v in lay.items():
            if isinstance(v, tuple):
                off, nb, dt, shape = v
                self._hv[name] = self._host[off:off + nb].view(dt).reshape(shape)
                self._dv[name] = self.dev[off:off + nb].view(self._TORCH[dt]).view(shape)
        self._spans = {"pixels": (self.pin[:key[2]], self.dev[:key[2]]),
                       "params": (self.pin[params_begin:lay["in_end"]], self.dev[params_begin:lay["in_end"]]),
                       "out": (self.pin[lay["out_begin"]:pos], self.dev[lay["out_begin"]:pos])}
        return lay

    def host(self, name) -> np.ndarray:
        return self._hv[name]

    def crops(self, n: int, hw) -> torch.Tensor:
        """The frame's warped crops: a device buffer kept per (n, size) instead of an allocation per frame."""
        key = (n, int(hw[0]), int(hw[1]))
        if getattr(self, "_crops_key", None) != key:
            self._crops = torch.empty(key, dtype=torch.float32, device=self.device)
            self._crops_key = key
        return self._crops

    def devt(self, name) -> torch.Tensor:
        return self._dv[name]

    def upload(self, span: str):
        """Enqueue the host-to-device copy of one span: "pixels" first, so that it travels while the host still
        packs the camera rows, then "params" (which ends with the zeroed status words)."""
        pin, dev = self._spans[span]
        dev.copy_(pin, non_blocking=True)

    def download(self):
        """Enqueue the copy of the output sections and wait for it: the frame's ONE host synchronisation."""
        pin, dev = self._spans["out"]
        pin.copy_(dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()


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
        self._io = _FrameIO(getattr(model, "device", None) or "cuda")
        self._hm_key: Optional[HandModel] = None     # the hand model whose device copies are cached below
        self._hm_dev: Optional[HandModel] = None
        self._skel: Optional[InputSkeletonData] = None
        self._src_rows = {}                          # id(camera) -> (camera, packed row): rig cameras are constant

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
        frame_data, frame_desc, skeleton, hands = self._make_inputs(sample, hand_model, crop_cameras)
        out = self._model.regress_pose_use_skeleton(frame_data, frame_desc, skeleton, required_rows=int(hands.max()) + 1,
                                                    out=self._out_views(False))
        return self._finish(out, hands, crop_cameras, hand_model)

    def track_frame_and_calibrate_scale(self, sample: InputFrame,
                                        crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]) -> TrackingResult:
        if not crop_cameras:
            self.reset_history()
            return TrackingResult()
        frame_data, frame_desc, _, hands = self._make_inputs(sample, None, crop_cameras)
        # umetrack_model.py:224-229; checked on the host copy of sample_range so that no device value is read back
        assert bool((np.diff(self._io.host("sr"), axis=1) != 1).all()), \
            "Unsupported: found single-view samples when calibration scale"
        out = self._model.regress_pose_pred_skel_scale(frame_data, frame_desc, required_rows=int(hands.max()) + 1,
                                                       out=self._out_views(True), views_checked=True)
        return self._finish(out, hands, crop_cameras, None)

    def _device_hand_model(self, hand_model_mm: HandModel):
        """Skeleton (metres) and FK model (mm) of `hand_model_mm` on the GPU; constant per recording, so they are
        uploaded when the caller's hand model changes, not once per frame (tracker.py:224-237 re-sends them)."""
        if self._hm_key is not hand_model_mm:
            hm = scaled_hand_model(hand_model_mm, MM_TO_M)
            dev = self._io.device
            self._skel = InputSkeletonData(joint_rotation_axes=hm.joint_rotation_axes.float().to(dev),
                                           joint_rest_positions=hm.joint_rest_positions.float().to(dev))
            self._hm_dev = hand_model_to(hand_model_mm, dev)
            self._hm_key = hand_model_mm
        return self._skel, self._hm_dev

    def _src_row(self, cam) -> np.ndarray:
        hit = self._src_rows.get(id(cam))
        if hit is None or hit[0] is not cam:
            if len(self._src_rows) > 64:
                self._src_rows.clear()
            hit = (cam, _pack_src(cam))
            self._src_rows[id(cam)] = hit
        return hit[1]

    def _make_inputs(self, sample: InputFrame, hand_model_mm: Optional[HandModel],
                     crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]]):
        """tracker.py:194-247.  All crops of the frame are warped in one kernel launch; every input reaches the GPU
        in one copy from a pinned block (_FrameIO)."""
        pairs, ranges, hands = [], [], []
        for hand_idx, per_hand in crop_cameras.items():
            first = len(pairs)
            for cam_idx, crop_cam in per_hand.items():
                pairs.append((cam_idx, crop_cam))
            if len(pairs) > first:
                hands.append(hand_idx)
                ranges.append((first, len(pairs)))
        hands = np.asarray(hands)
        n, bs = len(pairs), len(hands)
        used = sorted({vi for vi, _ in pairs})
        for vi in used:
            _check_view(vi, sample.views[vi].image)
        io = self._io
        parent = _common_parent([sample.views[vi].image for vi in used])
        start = {}                                               # view -> (byte offset of its first pixel, row pitch)
        if parent is not None:
            # the views are column blocks of one frame (video_pose_data.py:123-125 slices them so): ONE copy of that
            # frame into the pinned block - a single memcpy when they tile whole rows - instead of a strided copy per view
            # (1.2 MB: 160 -> 110 us of host time per frame)
            p_addr = parent.__array_interface__["data"][0]
            for vi in used:
                start[vi] = (sample.views[vi].image.__array_interface__["data"][0] - p_addr, parent.shape[1])
            io.plan(n, bs, _align(parent.size, 16))
            np.copyto(io.host("pixels")[:parent.size].reshape(parent.shape), parent)
        else:
            pos = 0
            for vi in used:
                img = sample.views[vi].image
                start[vi] = (pos, img.shape[1])
                pos += _align(img.size, 16)
            io.plan(n, bs, pos)
            px = io.host("pixels")
            for vi in used:
                img = sample.views[vi].image
                np.copyto(px[start[vi][0]:start[vi][0] + img.size].reshape(img.shape), img)
        io.upload("pixels")
        hv = io._hv
        src, dst, K, E = hv["src"], hv["dst"], hv["K"], hv["E"]
        c2w = np.stack([crop_cam.camera_to_world_xf for _, crop_cam in pairs]).astype(np.float64, copy=False)
        w2e = np.linalg.inv(c2w)                                 # tracker.py:213-215, all crops of the frame at once
        w2e[:, :3, 3] *= MM_TO_M
        E[:] = w2e.reshape(n, 16)
        dst[:, 4:] = c2w.reshape(n, 16)
        fc = np.empty((n, 4))                                    # fx, fy, cx, cy of every crop camera
        for k, (vi, crop_cam) in enumerate(pairs):
            src[k] = self._src_row(sample.views[vi].camera)
            f = crop_cam.f
            fc[k] = (f, f, crop_cam.c[0], crop_cam.c[1]) if np.ndim(f) == 0 else (f[0], f[1], crop_cam.c[0], crop_cam.c[1])
        dst[:, :4] = fc
        K[:] = 0                                                 # uv_to_window_matrix, camera.py:323-327
        K[:, 0], K[:, 4], K[:, 2], K[:, 5], K[:, 8] = fc[:, 0], fc[:, 1], fc[:, 2], fc[:, 3], 1
        hv["off"][:, 0] = [start[vi][0] for vi, _ in pairs]
        hv["pitch"][:, 0] = [start[vi][1] for vi, _ in pairs]
        hv["sr"][:] = ranges
        hv["midx"][:, 0] = hands
        hv["hidx"][:, 0] = hands
        hv["use"][:, 0] = self._valid_tracking_history[hands]    # enable_memory is stored but unused, as in the reference
        hv["status"][:] = 0
        io.upload("params")
        crops = gpu_warp.warp_crops(io.devt("pixels"), io.devt("off"), io.devt("pitch"), io.devt("src"), io.devt("dst"),
                                    tuple(int(v) for v in self._input_size), self._warp_mode, out=io.crops(n, self._input_size))
        frame_data = InputFrameData(left_images=crops, intrinsics=io.devt("K").view(n, 3, 3),
                                    extrinsics_xf=io.devt("E").view(n, 4, 4))
        frame_desc = InputFrameDesc(sample_range=io.devt("sr"), memory_idx=io.devt("midx").view(bs),
                                    use_memory=io.devt("use").view(bs).view(torch.bool), hand_idx=io.devt("hidx").view(bs))
        skeleton = None
        if hand_model_mm is not None:
            skeleton, _ = self._device_hand_model(hand_model_mm)
        return frame_data, frame_desc, skeleton, hands

    def _out_views(self, with_scales: bool) -> RegressorOutput:
        io = self._io
        bs = io.cap[1]
        return RegressorOutput(joint_angles=io.devt("angles"), wrist_xfs=io.devt("wrist").view(bs, 4, 4),
                               skel_scales=io.devt("scales").view(bs) if with_scales else None,
                               landmark_uncertainty_sigmas=io.devt("sigmas"))

    def _finish(self, out: RegressorOutput, hands: np.ndarray,
                crop_cameras: Dict[int, Dict[int, camera.PinholePlaneCameraModel]],
                hand_model_mm: Optional[HandModel]) -> TrackingResult:
        """tracker.py:156-161, 249-291: outputs to the host (one copy, with the landmarks and the model's status
        word in it), metres -> millimetres, per-hand dicts, tracking history update."""
        io = self._io
        bs = len(hands)
        status_dev = io.devt("status").view(-1)
        if hand_model_mm is not None:       # the landmarks the eval loop asks for next (perspective_crop.py:40-51)
            _, hm_dev = self._device_hand_model(hand_model_mm)
            skin_landmarks(hm_dev, out.joint_angles, out.wrist_xfs, hand_idx=io.devt("hidx").view(bs), mirror_right=True,
                           wrist_translation_scale=M_TO_MM, out=io.devt("lm").view(bs, 21, 3), status=status_dev[1:2])
        self._model.status_to(status_dev[0:1])
        io.download()
        bits = int(io.host("status")[0, 0]) | int(io.host("status")[0, 1])
        raising = bits & ~nat.STATUS_NONFINITE_OUTPUT
        if raising:                         # where the reference raises inside the forward pass
            raise nat.UmtError("; ".join(msg for bit, msg in nat.STATUS_BITS.items() if raising & bit))
        if bits & nat.STATUS_NONFINITE_OUTPUT:      # the reference returns the NaN pose and goes on
            logger.warning("non-finite regressor output in this frame (NaN pose returned)")
        angles = io.host("angles").copy()
        wrists = io.host("wrist").reshape(bs, 4, 4).copy()
        wrists[..., :3, 3] *= M_TO_MM
        scales = io.host("scales").reshape(bs).copy() if out.skel_scales is not None else None
        lms = io.host("lm").reshape(bs, 21, 3).copy() if hand_model_mm is not None else None
        poses, views, pred_scales = {}, {}, {}
        for row, hand_idx in enumerate(hands):
            pose = TrackedHandPose(joint_angles=angles[row], wrist_xform=wrists[row], hand_confidence=1.0)
            if lms is not None:
                pose.landmarks_mm, pose.landmarks_model, pose.landmarks_hand_idx = lms[row], hand_model_mm, int(hand_idx)
            poses[hand_idx] = pose
            views[hand_idx] = len(crop_cameras[hand_idx])
            if scales is not None:
                pred_scales[hand_idx] = scales[row]
        for hand_idx in range(NUM_HANDS):
            self._valid_tracking_history[hand_idx] = hand_idx in poses
        return TrackingResult(hand_poses=poses, num_views=views, predicted_scales=pred_scales)
