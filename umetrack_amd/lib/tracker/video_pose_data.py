"""Raw-data recordings: video of 4-view-wide mono frames + a label JSON.

Names follow lib/tracker/video_pose_data.py:23-151.  The label side (cameras, camera angles,
per-frame camera_to_world, hand model, joint angles, wrist transforms, confidences) and the
frame -> (InputFrame, gt_tracking) logic are complete.  The pixel side has two sources:

  * ``VideoStream``    - mp4 through PyAV, as in the reference.  PyAV is not installed in this
                         image, so this class raises ImportError when opened ("parity unpinned":
                         the decode itself is third-party code on both sides);
  * ``RawMonoStream``  - a headerless file of uint8 frames (H x views*W each), the same pixels
                         without a codec, written by ``write_recording``; used by the tests and
                         for synthetic recordings on disk.

``SyncedImagePoseStream(path)`` picks the source by extension (.mp4 -> VideoStream,
.mono -> RawMonoStream) and reads ``path[:-4] + ".json"`` (video_pose_data.py:98) or, for
.mono, ``path[:-5] + ".json"``.
"""
import json
import os
from dataclasses import dataclass
from typing import Iterator, List

import numpy as np

from ..common.camera import CameraModel, read_camera_from_json
from ..common.hand import HandModel, load_hand_model_from_dict
from .tracker import InputFrame, ViewData
from .tracking_result import SingleHandPose


@dataclass
class HandPoseLabels:
    cameras: List[CameraModel]
    camera_angles: List[float]
    camera_to_world_transforms: np.ndarray     # (frames, cameras, 4, 4)
    hand_model: HandModel
    joint_angles: np.ndarray                   # (frames, 2, 22)
    wrist_transforms: np.ndarray               # (frames, 2, 4, 4)
    hand_confidences: np.ndarray               # (frames, 2)

    def __len__(self):
        return len(self.joint_angles)


class VideoStream:
    """First video stream of an mp4, one mono frame (H, views*W) per iteration (video_pose_data.py:37-55)."""

    def __init__(self, data_path: str):
        self._data_path = data_path

    @staticmethod
    def _av():
        try:
            import av
        except ImportError as e:          # not in this image; never stubbed
            raise ImportError("reading .mp4 recordings needs PyAV (`av`), which is not installed; "
                              "convert the recording to .mono with any decoder, or install PyAV") from e
        return av

    def __len__(self) -> int:
        return self._av().open(self._data_path).streams.video[0].frames

    def __iter__(self) -> Iterator[np.ndarray]:
        container = self._av().open(self._data_path)
        stream = container.streams.video[0]
        for frame in container.decode(stream):
            yield np.array(frame.to_image())[..., 0]


class RawMonoStream:
    """Headerless uint8 frames, memory-mapped; the frame shape comes from the labels."""

    def __init__(self, data_path: str, height: int, width: int):
        self._shape = (height, width)
        nbytes = os.path.getsize(data_path)
        if nbytes % (height * width):
            raise ValueError(f"{data_path}: {nbytes} bytes is not a whole number of {height}x{width} frames")
        self._frames = np.memmap(data_path, dtype=np.uint8, mode="r", shape=(nbytes // (height * width), height, width))

    def __len__(self) -> int:
        return self._frames.shape[0]

    def __iter__(self) -> Iterator[np.ndarray]:
        for f in self._frames:
            yield f

    @property
    def frames(self) -> np.ndarray:
        """All frames (n, H, views*W) as one zero-copy array: what the bulk GPU path uploads."""
        return self._frames


def _load_hand_pose_labels(p: str) -> HandPoseLabels:
    with open(p, "rb") as fh:
        labels = json.load(fh)
    return HandPoseLabels(cameras=[read_camera_from_json(c) for c in labels["cameras"]],
                          camera_angles=labels["camera_angles"],
                          camera_to_world_transforms=np.array(labels["camera_to_world_transforms"]),
                          hand_model=load_hand_model_from_dict(labels["hand_model"]),
                          joint_angles=np.array(labels["joint_angles"]),
                          wrist_transforms=np.array(labels["wrist_transforms"]),
                          hand_confidences=np.array(labels["hand_confidences"]))


class SyncedImagePoseStream:
    """Iterates (InputFrame, {hand_idx: SingleHandPose}) like video_pose_data.py:96-151."""

    def __init__(self, data_path: str):
        raw = data_path.endswith(".mono")
        self._hand_pose_labels = _load_hand_pose_labels(data_path[:-5 if raw else -4] + ".json")
        cams = self._hand_pose_labels.cameras
        if raw:
            self._image_stream = RawMonoStream(data_path, cams[0].height, sum(c.width for c in cams))
        else:
            self._image_stream = VideoStream(data_path)
        assert len(self._hand_pose_labels) == len(self._image_stream)

    def __len__(self) -> int:
        return len(self._image_stream)

    @property
    def labels(self) -> HandPoseLabels:
        return self._hand_pose_labels

    @property
    def camera_angles(self) -> List[float]:
        return self._hand_pose_labels.camera_angles

    @property
    def hand_model(self) -> HandModel:
        return self._hand_pose_labels.hand_model

    def __iter__(self):
        lab = self._hand_pose_labels
        n_cam = len(lab.cameras)
        for frame_idx, raw_mono in enumerate(self._image_stream):
            gt_tracking = {}
            for hand_idx in range(2):
                conf = lab.hand_confidences[frame_idx, hand_idx]
                if conf > 0:
                    gt_tracking[hand_idx] = SingleHandPose(joint_angles=lab.joint_angles[frame_idx, hand_idx],
                                                           wrist_xform=lab.wrist_transforms[frame_idx, hand_idx],
                                                           hand_confidence=conf)
            views_wide = raw_mono.reshape(raw_mono.shape[0], n_cam, -1)          # views side by side
            if lab.camera_to_world_transforms[frame_idx].sum() == 0:
                assert not gt_tracking, "Cameras are not tracked, expecting no ground truth tracking!"
            views = [ViewData(image=views_wide[:, c, :],
                              camera=lab.cameras[c].copy(camera_to_world_xf=lab.camera_to_world_transforms[frame_idx, c]),
                              camera_angle=lab.camera_angles[c]) for c in range(n_cam)]
            yield InputFrame(views=views), gt_tracking


# ---- writer (no counterpart in the reference: its recordings come from the dataset download) ----
def camera_to_json(cam: CameraModel) -> dict:
    """Inverse of read_camera_from_json (camera.py:334-355)."""
    name = {"Fisheye62CameraModel": "FishEye62", "PinholePlaneCameraModel": "PinholePlane"}[type(cam).__name__]
    js = {"DistortionModel": name, "ImageSizeX": int(cam.width), "ImageSizeY": int(cam.height),
          "fx": float(cam.f[0]), "fy": float(cam.f[1]), "cx": float(cam.c[0]), "cy": float(cam.c[1])}
    js.update({k: float(v) for k, v in cam.distort._asdict().items()})
    return {"Camera": js}


def write_recording(path_mono: str, frames: np.ndarray, cameras: List[CameraModel], camera_angles, hand_model: HandModel,
                    joint_angles, wrist_transforms, hand_confidences, camera_to_world_transforms=None) -> None:
    """frames (n, H, views*W) uint8 -> ``path_mono`` (.mono) and the label JSON next to it."""
    assert path_mono.endswith(".mono")
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n = frames.shape[0]
    if camera_to_world_transforms is None:
        camera_to_world_transforms = np.broadcast_to(np.stack([c.camera_to_world_xf for c in cameras]), (n, len(cameras), 4, 4))
    frames.tofile(path_mono)
    hm = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in hand_model._asdict().items() if v is not None}
    labels = {"cameras": [camera_to_json(c) for c in cameras], "camera_angles": [float(a) for a in camera_angles],
              "camera_to_world_transforms": np.asarray(camera_to_world_transforms).tolist(), "hand_model": hm,
              "joint_angles": np.asarray(joint_angles).tolist(), "wrist_transforms": np.asarray(wrist_transforms).tolist(),
              "hand_confidences": np.asarray(hand_confidences).tolist()}
    with open(path_mono[:-5] + ".json", "w") as fh:
        json.dump(labels, fh)
