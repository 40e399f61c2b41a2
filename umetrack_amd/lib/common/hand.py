"""Hand model container and helpers; same names as lib/common/hand.py:12-134."""
import json
import os
from typing import NamedTuple, Optional

import torch

NUM_HANDS = 2
NUM_LANDMARKS_PER_HAND = 21
NUM_FINGERTIPS_PER_HAND = 5
NUM_JOINTS_PER_HAND = 22
LEFT_HAND_INDEX = 0
RIGHT_HAND_INDEX = 1
NUM_DIGITS = 5
NUM_JOINT_FRAMES = 1 + 1 + 3 * 5
DOF_PER_FINGER = 4


class HandModel(NamedTuple):
    joint_rotation_axes: torch.Tensor
    joint_rest_positions: torch.Tensor
    joint_frame_index: torch.Tensor
    joint_parent: torch.Tensor
    joint_first_child: torch.Tensor
    joint_next_sibling: torch.Tensor
    landmark_rest_positions: torch.Tensor
    landmark_rest_bone_weights: torch.Tensor
    landmark_rest_bone_indices: torch.Tensor
    hand_scale: Optional[torch.Tensor]
    mesh_vertices: Optional[torch.Tensor] = None
    mesh_triangles: Optional[torch.Tensor] = None
    dense_bone_weights: Optional[torch.Tensor] = None
    joint_limits: Optional[torch.Tensor] = None


def hand_model_to(hand_model: HandModel, device) -> HandModel:
    """The model with its tensor fields resident on ``device`` in the dtypes the kernels take (float32, bone indices
    int32), so that `skin_landmarks` / `crop_points_batched` find them in place instead of copying 5 small host tensors
    per call (pageable host-to-device copies synchronise the stream).  No counterpart in the reference, whose
    `skin_landmarks` runs where its inputs live."""
    def move(k, v):
        if not isinstance(v, torch.Tensor):
            return v
        if k == "landmark_rest_bone_indices":
            return v.to(device=device, dtype=torch.int32).contiguous()
        return v.to(device=device, dtype=torch.float32 if v.is_floating_point() else v.dtype).contiguous()
    return HandModel(**{k: move(k, v) for k, v in hand_model._asdict().items()})


def load_hand_model_from_dict(d) -> HandModel:
    """lib/tracker/video_pose_data.py:63-72: lists become float tensors, scalars stay."""
    return HandModel(**{k: (torch.Tensor(v) if isinstance(v, list) else v) for k, v in d.items()})


def load_generic_hand_model() -> HandModel:
    """The hot-path fields of dataset/generic_hand_model.json (mesh omitted), units mm."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                        "data", "generic_hand_model_small.json")
    with open(path) as fh:
        return load_hand_model_from_dict(json.load(fh))


def _per_sample(t: torch.Tensor, factor) -> torch.Tensor:
    return t * factor[..., None, None]


def scaled_hand_model(hand: HandModel, multiplier) -> HandModel:
    """hand.py:65-98: scale rest positions, landmarks and mesh; everything else is shared."""
    ref = hand.joint_rest_positions
    factor = torch.ones(ref.shape[:-2], dtype=ref.dtype, device=ref.device) * multiplier
    return hand._replace(
        joint_rest_positions=_per_sample(hand.joint_rest_positions, factor),
        landmark_rest_positions=_per_sample(hand.landmark_rest_positions, factor),
        mesh_vertices=None if hand.mesh_vertices is None else _per_sample(hand.mesh_vertices, factor))


def mirrored_hand_model(hand: HandModel, to_mirror: torch.Tensor) -> HandModel:
    """hand.py:101-134: flip x of positions and y,z of axes where `to_mirror` is set."""
    sel = to_mirror.reshape(-1).bool()
    lead = to_mirror.dim()

    def flip(t, cols):
        out = t.clone()
        flat = out.reshape((-1,) + tuple(t.shape[lead:])) if lead else out.unsqueeze(0)
        sign = torch.ones(t.shape[-1], dtype=t.dtype, device=t.device)
        sign[cols] = -1
        flat[sel] = flat[sel] * sign
        return flat.reshape(t.shape)

    return hand._replace(joint_rotation_axes=flip(hand.joint_rotation_axes, slice(1, None)),
                         joint_rest_positions=flip(hand.joint_rest_positions, slice(0, 1)),
                         landmark_rest_positions=flip(hand.landmark_rest_positions, slice(0, 1)),
                         mesh_vertices=None if hand.mesh_vertices is None else hand.mesh_vertices.clone())
