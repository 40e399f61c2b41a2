"""oracle/fk.py: pinned by the reference's hand-model data; rotation formula unpinned."""
import numpy as np
from scipy.spatial.transform import Rotation

from oracle import fk


def test_rest_pose_reproduces_reference_landmark_rest_positions(hand_model_np):
    lm = fk.skin_landmarks(hand_model_np, np.zeros(22, np.float32), np.eye(4, dtype=np.float32))
    np.testing.assert_allclose(lm, hand_model_np["landmark_rest_positions"], atol=2e-5)   # mm


def test_wrist_transform_moves_landmarks_rigidly(hand_model_np):
    rng = np.random.default_rng(2)
    W = np.eye(4)
    W[:3, :3] = Rotation.from_rotvec(rng.normal(size=3)).as_matrix()
    W[:3, 3] = rng.normal(size=3) * 100
    ja = np.zeros(22)
    lm = fk.skin_landmarks(hand_model_np, ja, W, np.float64)
    exp = hand_model_np["landmark_rest_positions"] @ W[:3, :3].T + W[:3, 3]
    np.testing.assert_allclose(lm, exp, atol=1e-9)


def test_so3_exp_against_scipy_and_small_angle_clamp():
    rng = np.random.default_rng(3)
    v = rng.normal(size=(50, 3))
    np.testing.assert_allclose(fk.so3_exp(v), Rotation.from_rotvec(v).as_matrix(), atol=1e-12)
    tiny = np.array([[1e-3, 0, 0]])
    R = fk.so3_exp(tiny)
    # below the clamp (theta^2 < 1e-4) the published formula is evaluated at theta = 0.01
    assert abs(R[0, 2, 1] - 1e-3 * np.sin(0.01) / 0.01) < 1e-15


def test_fp32_against_fp64_and_batch_shapes(hand_model_np):
    rng = np.random.default_rng(4)
    lim = hand_model_np["joint_limits"]
    ja = rng.uniform(lim[:, 0], lim[:, 1], size=(3, 5, 22))
    W = np.tile(np.eye(4), (3, 5, 1, 1))
    W[..., :3, 3] = rng.normal(size=(3, 5, 3)) * 200
    hm = {k: (np.broadcast_to(v, (3, 5) + v.shape).copy() if v.ndim >= 1 else v) for k, v in hand_model_np.items()}
    a = fk.skin_landmarks(hm, ja, W, np.float32)
    b = fk.skin_landmarks(hm, ja, W, np.float64)
    c = fk.skin_landmarks(hand_model_np, ja, W, np.float64)          # un-batched model broadcasts
    assert a.shape == (3, 5, 21, 3)
    np.testing.assert_allclose(a, b, atol=2e-3)       # mm, fp32 rounding at ~500 mm
    np.testing.assert_allclose(b, c, atol=1e-12)
    right = fk.landmarks_from_pose(hand_model_np, ja[0, 0], W[0, 0], 1, np.float64)
    left = fk.landmarks_from_pose(hand_model_np, ja[0, 0], W[0, 0], 0, np.float64)
    assert not np.allclose(right, left)
