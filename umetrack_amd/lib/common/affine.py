"""Small rigid-transform helpers (host, numpy float64); names follow lib/common/affine.py:11-66."""
import numpy as np


def transform_vec3(m: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Rotate points by the upper-left 3x3 of m (one matrix, or one per point)."""
    if m.ndim == 2:
        return (v.reshape(-1, 3) @ m[:3, :3].T).reshape(v.shape)
    return np.einsum("...ij,...j->...i", m[..., :3, :3], v)


def transform3(m: np.ndarray, v: np.ndarray) -> np.ndarray:
    return transform_vec3(m, v) + m[..., :3, 3]


def normalized(v: np.ndarray, axis: int = -1, eps: float = 5.43e-20) -> np.ndarray:
    return v / np.maximum(eps, np.sqrt((v * v).sum(axis=axis, keepdims=True)))


def skew_matrix(v: np.ndarray) -> np.ndarray:
    x, y, z = v
    return np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=v.dtype)


def from_two_vectors(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Rotation taking direction a to direction b (affine.py:34-44)."""
    a, b = normalized(a), normalized(b)
    axis = np.cross(a, b)
    k = skew_matrix(axis)
    s2 = float(axis @ axis)
    return np.eye(3) + k + (k @ k) * ((1 - float(a @ b)) / max(s2, 1e-15))


def rot_z_deg(angle_deg: float) -> np.ndarray:
    """scipy Rotation.from_euler('z', angle, degrees=True).as_matrix() (affine.py:63)."""
    t = np.deg2rad(angle_deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def make_look_at_matrix(orig_world_to_eye: np.ndarray, center: np.ndarray, camera_angle: float = 0) -> np.ndarray:
    """world->eye of a camera at the same position looking at `center`, rolled by camera_angle
    (affine.py:47-66)."""
    local = transform3(orig_world_to_eye, center)
    delta = from_two_vectors(np.array([0, 0, 1], dtype=center.dtype), local / np.linalg.norm(local))
    eye_to_world = np.linalg.inv(orig_world_to_eye)
    new = eye_to_world.copy()
    new[:3, :3] = eye_to_world[:3, :3] @ delta @ rot_z_deg(camera_angle)
    return np.linalg.inv(new)
