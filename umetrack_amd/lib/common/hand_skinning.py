"""skin_landmarks with the reference's signature (lib/common/hand_skinning.py:189-209).

The forward-kinematics chain and the blend run in one HIP kernel
(csrc/fk.hip, umt_fk_landmarks); this wrapper only flattens leading
dimensions.  There is no CPU implementation in this package.
"""
from typing import Optional

import torch

from ... import _native as nat
from .hand import HandModel


def skin_landmarks(hand_model: HandModel, joint_angles: torch.Tensor, wrist_transforms: torch.Tensor,
                   hand_idx: Optional[torch.Tensor] = None, mirror_right: bool = False,
                   wrist_translation_scale: float = 1.0, out: Optional[torch.Tensor] = None,
                   status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """joint_angles [...,22], wrist_transforms [...,4,4] -> landmarks [...,21,3] (float32, on the GPU).

    Model fields either carry the same leading dims or none (one shared model).
    ``hand_idx`` / ``mirror_right`` fold in the right-hand mirror of
    landmarks_from_hand_pose (perspective_crop.py:46-49); ``wrist_translation_scale``
    folds in the m->mm step of the tracker (tracker.py:258).

    Bone indices: the reference scatters the blend weights through a flattened index
    (hand_skinning.py:87-94), so an index outside [0, 16] corrupts a neighbouring landmark or raises
    IndexError.  Here a host-resident index tensor is checked before the launch (IndexError); for a
    device-resident one the kernel writes NaN landmarks and ORs UMT_STATUS_BAD_BONE_INDEX (16) into
    ``status`` (int32[1] on the device) when one is given.
    """
    dev = nat.require_gpu(joint_angles.device if joint_angles.is_cuda else None)
    idx_t, w_t = hand_model.landmark_rest_bone_indices, hand_model.landmark_rest_bone_weights
    if not (isinstance(idx_t, torch.Tensor) and idx_t.is_cuda):        # 63 integers per model: checked on the host
        ih, wh = torch.as_tensor(idx_t), torch.as_tensor(w_t).cpu()
        bad = ((ih < 0) | (ih > 16)) & (wh != 0)
        if bool(bad.any()):
            raise IndexError(f"landmark_rest_bone_indices: index {int(ih[bad][0])} is out of bounds for 17 bone frames")
    lead = tuple(joint_angles.shape[:-1])
    f32 = torch.float32
    ja = nat.dev_tensor(joint_angles, f32, dev).reshape(-1, 22)
    b = ja.shape[0]
    wr = nat.dev_tensor(wrist_transforms, f32, dev).reshape(-1, 4, 4)
    assert wr.shape[0] == b, f"Leading dimensions do not match, got {lead} and {tuple(wrist_transforms.shape[:-2])}"

    def field(t, tail, dtype):
        t = nat.dev_tensor(t, dtype, dev)
        shared = t.dim() == len(tail)
        assert shared or tuple(t.shape[:-len(tail)]) == lead, \
            f"Leading dimensions do not match, got {lead} and {tuple(t.shape[:-len(tail)])}"
        return t.reshape((-1,) + tail), shared

    axes, s0 = field(hand_model.joint_rotation_axes, (22, 3), f32)
    rest, s1 = field(hand_model.joint_rest_positions, (22, 3), f32)
    lm, s2 = field(hand_model.landmark_rest_positions, (21, 3), f32)
    lw, s3 = field(hand_model.landmark_rest_bone_weights, (21, 3), f32)
    li, s4 = field(hand_model.landmark_rest_bone_indices, (21, 3), torch.int32)
    flags = {s0, s1, s2, s3, s4}
    if len(flags) != 1:       # mixed: expand the shared ones
        axes, rest, lm, lw, li = (t.expand(b, *t.shape[1:]).contiguous() if t.shape[0] == 1 else t
                                  for t in (axes, rest, lm, lw, li))
    n_model = axes.shape[0]
    hidx = None if hand_idx is None else nat.dev_tensor(hand_idx, torch.int64, dev).reshape(-1)
    if out is None:
        out = torch.empty(b, 21, 3, dtype=f32, device=dev)
    else:
        assert out.is_contiguous() and out.dtype == f32 and out.numel() == b * 63 and out.device == dev
    with torch.cuda.device(dev):
        nat.check(nat.lib().umt_fk_landmarks(nat.ptr(ja), nat.ptr(wr), nat.ptr(axes), nat.ptr(rest), nat.ptr(lm),
                                             nat.ptr(lw), nat.ptr(li), n_model, nat.ptr(hidx), int(mirror_right),
                                             float(wrist_translation_scale), b, nat.ptr(out), nat.ptr(status),
                                             nat.stream_ptr(dev)),
                  "umt_fk_landmarks")
    return out.reshape(lead + (21, 3))
