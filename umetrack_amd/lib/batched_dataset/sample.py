"""One torch_data sequence as typed arrays; field names of lib/batched_dataset/sample.py:16-53."""
from dataclasses import dataclass
from typing import Any, Dict

import numpy as np
import torch

from ..common.hand import HandModel, scaled_hand_model


@dataclass
class RawSample:
    images: np.ndarray              # (T, views, H, W)
    extrinsics: np.ndarray          # (T, views, 4, 4) world->eye, mm
    intrinsics: np.ndarray          # (T, views, 3, 3)
    enclosing_points: np.ndarray    # (T, P, 3) crop points, mm
    hand: np.ndarray                # (T,) 0 = left, 1 = right
    hand_model: HandModel           # ground-truth skeleton
    wrist: np.ndarray               # (T, 4, 4)
    joint_angles: np.ndarray        # (T, 22)
    solved_wrist_xfs: np.ndarray    # pose solved on the generic skeleton
    solved_joint_angles: np.ndarray
    generic_hand_model: HandModel
    pinch: np.ndarray

    def scaled(self, factor: float) -> None:
        """mm -> m in place (sample.py:33-39)."""
        self.extrinsics[..., :3, 3] *= factor
        self.enclosing_points *= factor
        self.wrist[..., :3, 3] *= factor
        self.solved_wrist_xfs[..., :3, 3] *= factor
        self.hand_model = scaled_hand_model(self.hand_model, factor)
        self.generic_hand_model = scaled_hand_model(self.generic_hand_model, factor)


def parse_raw_buffers(mono: np.ndarray, labels: Dict[str, Any]) -> RawSample:
    """Decoded ``mono`` frame + ``labels`` msgpack document -> RawSample (sample.py:42-53): hand models
    become tensors, every other label a float32 array."""
    typed: Dict[str, Any] = {}
    for name, value in labels.items():
        if "hand_model" in name:
            typed[name] = HandModel(**{k: torch.tensor(v) for k, v in value.items()})
        else:
            typed[name] = np.array(value, dtype=np.float32)
    return RawSample(images=mono, **typed)
