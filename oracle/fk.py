"""CPU restatement of hand forward kinematics + linear blend skinning (row A14).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
lib/common/hand_skinning.py:17-209 and lib/tracker/perspective_crop.py:40-51.

Pinning: the reference's ``hand_skinning`` cannot be imported here (it needs
pytorch3d, which is not in this image), and the reference ships no FK output.
What *is* pinned by reference-held data: at zero joint angles and identity wrist
the landmarks equal ``landmark_rest_positions`` of dataset/generic_hand_model.json
(tests/test_oracle_fk.py).  PARITY UNPINNED for the rotation itself:
``so3_exp_map`` is restated from pytorch3d's published formula (Rodrigues with
the squared angle clamped below at 1e-4), ``R = I + sin(t)/t K + (1-cos t)/t^2 K^2``.
"""
import numpy as np

NUM_FRAMES = 17          # hand.py:20: root + wrist + 3 per finger
SO3_EPS = 1e-4           # pytorch3d so3_exp_map default eps


def so3_exp(log_rot: np.ndarray) -> np.ndarray:
    dt = log_rot.dtype
    nrm2 = (log_rot * log_rot).sum(-1)
    theta = np.sqrt(np.maximum(nrm2, dt.type(SO3_EPS)))
    inv = dt.type(1) / theta
    fac1 = inv * np.sin(theta)
    fac2 = inv * inv * (dt.type(1) - np.cos(theta))
    x, y, z = log_rot[..., 0], log_rot[..., 1], log_rot[..., 2]
    zero = np.zeros_like(x)
    K = np.stack([np.stack([zero, -z, y], -1), np.stack([z, zero, -x], -1),
                  np.stack([-y, x, zero], -1)], -2)
    return fac1[..., None, None] * K + fac2[..., None, None] * (K @ K) + np.eye(3, dtype=dt)


def skinning_frames(axes, rest, angles, wrist) -> np.ndarray:
    """hand_skinning.py:100-127: (B,17,4,4).  Frames 0,1 = wrist; per finger f the
    chain W.L0.L1, W.L0.L1.L2, W.L0.L1.L2.L3 (hand_skinning.py:17-32)."""
    B = angles.shape[0]
    dt = angles.dtype
    R = so3_exp(axes[:, :20] * angles[:, :20, None])
    L = np.tile(np.eye(4, dtype=dt), (B, 20, 1, 1))
    L[..., :3, :3] = R
    L[..., :3, 3] = rest[:, :20] - (R @ rest[:, :20, :, None])[..., 0]
    frames = [wrist, wrist]
    for f in range(5):
        T = wrist
        for j in range(4):
            T = T @ L[:, 4 * f + j]
            if j >= 1:
                frames.append(T)
    return np.stack(frames, 1)


def dense_weights(bone_idx, bone_w) -> np.ndarray:
    """hand_skinning.py:70-97: scatter (assignment, zero weights skipped) to (B,V,17)."""
    B, V, K = bone_idx.shape
    out = np.zeros((B, V, NUM_FRAMES), dtype=bone_w.dtype)
    for k in range(K):
        b, v = np.nonzero(bone_w[..., k] != 0)
        out[b, v, bone_idx[b, v, k].astype(np.int64)] = bone_w[b, v, k]
    return out


def skin_landmarks(model: dict, joint_angles, wrist_xf, dtype=np.float32) -> np.ndarray:
    """hand_skinning.py:189-209 for inputs with arbitrary leading dims.

    ``model`` maps field name -> array; model fields either share the leading
    dims of ``joint_angles`` or have none (broadcast, as the tracker uses it).
    """
    ja = np.asarray(joint_angles, dtype=dtype)
    lead = ja.shape[:-1]
    B = int(np.prod(lead)) if lead else 1
    ja = ja.reshape(B, -1)
    W = np.asarray(wrist_xf, dtype=dtype).reshape(B, 4, 4)

    def field(name, tail):
        a = np.asarray(model[name])
        a = a.astype(dtype) if a.dtype.kind == "f" else a
        return np.broadcast_to(a.reshape((-1,) + tail), (B,) + tail) if a.ndim == len(tail) \
            else a.reshape((B,) + tail)

    axes, rest = field("joint_rotation_axes", (22, 3)), field("joint_rest_positions", (22, 3))
    lm = field("landmark_rest_positions", (21, 3))
    bw = field("landmark_rest_bone_weights", (21, 3))
    bi = field("landmark_rest_bone_indices", (21, 3))
    frames = skinning_frames(axes, rest, ja, W)
    wts = dense_weights(bi, bw)
    homo = np.concatenate([lm, np.ones((B, 21, 1), dtype=dtype)], -1)
    pts = np.einsum("bfij,bvj->bvfi", frames, homo)            # (B,V,17,4)
    out = (pts * wts[..., None]).sum(2)[..., :3]
    return out.reshape(lead + (21, 3)).astype(dtype)


def landmarks_from_pose(model: dict, joint_angles, wrist_xf, hand_idx: int, dtype=np.float32):
    """perspective_crop.py:40-51: right hands negate column 0 of the wrist transform."""
    xf = np.array(wrist_xf, dtype=np.float64, copy=True)
    if hand_idx == 1:
        xf[:, 0] *= -1
    return skin_landmarks(model, np.asarray(joint_angles), xf, dtype)
