"""Result containers, same fields as lib/tracker/tracking_result.py:14-30."""
from typing import Dict, NamedTuple

import numpy as np

from ..common.hand import NUM_JOINTS_PER_HAND


class SingleHandPose(NamedTuple):
    joint_angles: np.ndarray = np.zeros(NUM_JOINTS_PER_HAND, dtype=np.float32)
    wrist_xform: np.ndarray = np.eye(4, dtype=np.float32)
    hand_confidence: float = 1.0


class TrackingResult(NamedTuple):
    hand_poses: Dict[int, SingleHandPose] = {}
    num_views: Dict[int, int] = {}
    predicted_scales: Dict[int, float] = {}
