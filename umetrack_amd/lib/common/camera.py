"""Camera models on the host (numpy float64): Pinhole and Fisheye62.

Class and method names follow lib/common/camera.py:164-355 so tracker code reads
the same; per-pixel projection for the warp itself runs on the GPU (csrc/warp.hip).
"""
import json
from typing import NamedTuple, Sequence

import numpy as np

from . import affine


class NoDistortion(NamedTuple):
    def evaluate(self, p: np.ndarray) -> np.ndarray:
        return p


class Fisheye62Distortion(NamedTuple):
    """6 radial + 2 tangential coefficients in the label-file order (camera.py:106-141)."""
    k1: float
    k2: float
    k3: float
    k4: float
    p1: float
    p2: float
    k5: float
    k6: float

    def evaluate(self, p: np.ndarray) -> np.ndarray:
        px, py = p[..., 0], p[..., 1]
        r2 = np.clip(px * px + py * py, -np.pi ** 2, np.pi ** 2)
        r4 = r2 * r2
        r6 = r2 * r4
        radial = (1 + self.k1 * r2 + self.k2 * r4 + self.k3 * r6 + self.k4 * (r4 * r4)
                  + self.k5 * (r4 * r6) + self.k6 * (r6 * r6))
        x, y = px * radial, py * radial
        xx, yy, xy = x * x, y * y, x * y
        rr = xx + yy
        return np.stack((x + (2 * self.p2 * xy + self.p1 * (rr + 2 * xx)),
                         y + (2 * self.p1 * xy + self.p2 * (rr + 2 * yy))), axis=-1)


class CameraModel:
    """width/height, f=(fx,fy), c=(cx,cy), distortion, camera_to_world_xf (4x4, mutable)."""
    distortion_model = NoDistortion

    def __init__(self, width, height, f, c, distort_coeffs, camera_to_world_xf=None):
        self.width, self.height = width, height
        self.f = tuple(np.broadcast_to(f, 2))
        self.c = tuple(c)
        self.camera_to_world_xf = np.eye(4) if camera_to_world_xf is None else camera_to_world_xf
        self.distort = distort_coeffs if hasattr(distort_coeffs, "evaluate") \
            else self.distortion_model(*distort_coeffs)

    def __repr__(self):
        return f"{type(self).__name__}({self.width}x{self.height}, f={self.f} c={self.c}"

    # -- to be provided by the projection mix of the subclass
    @staticmethod
    def project(v):
        raise NotImplementedError

    @staticmethod
    def unproject(p):
        raise NotImplementedError

    def copy(self, camera_to_world_xf=None):
        return self.crop(0, 0, self.width, self.height, camera_to_world_xf=camera_to_world_xf)

    def world_to_eye(self, v):
        xf = self.camera_to_world_xf
        return affine.transform_vec3(xf.T, v - xf[:3, 3])

    def eye_to_world(self, v):
        return affine.transform3(self.camera_to_world_xf, v)

    def eye_to_window(self, v):
        return self.distort.evaluate(self.project(v)) * self.f + self.c

    def window_to_eye(self, w):
        assert isinstance(self.distort, NoDistortion), "Only unprojection for NoDistortion camera is supported"
        return self.unproject((np.asarray(w) - self.c) / self.f)

    def crop(self, src_x, src_y, target_width, target_height, scale=1, camera_to_world_xf=None):
        return type(self)(target_width, target_height, np.asarray(self.f) * scale,
                          (np.array(self.c) - (src_x, src_y) + 0.5) * scale - 0.5, self.distort,
                          self.camera_to_world_xf if camera_to_world_xf is None else camera_to_world_xf)


class PinholePlaneCameraModel(CameraModel):
    distortion_model = NoDistortion

    @staticmethod
    def project(v):
        return v[..., :2] / v[..., 2, None]

    @staticmethod
    def unproject(p):
        ones = np.ones(p.shape[:-1] + (1,), dtype=p.dtype)
        return affine.normalized(np.concatenate((p, ones), axis=-1), axis=-1)

    def uv_to_window_matrix(self):
        return np.array([[self.f[0], 0, self.c[0]], [0, self.f[1], self.c[1]], [0, 0, 1]])


class Fisheye62CameraModel(CameraModel):
    distortion_model = Fisheye62Distortion

    @staticmethod
    def project(p, eps: float = 2.0 ** -128):
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        r = np.sqrt(x * x + y * y)
        s = np.arctan2(r, z) / np.maximum(r, eps)
        return np.stack((x * s, y * s), axis=-1)

    @staticmethod
    def unproject(uv):
        r = np.sqrt((uv * uv).sum(-1))
        s = np.sinc(r / np.pi)
        return np.stack((uv[..., 0] * s, uv[..., 1] * s, np.cos(r)), axis=-1)


def read_camera_from_json(js):
    """Label-file camera record -> camera object (camera.py:334-355)."""
    if isinstance(js, str):
        js = json.loads(js)
    js = js.get("Camera", js)
    cls = {"PinholePlane": PinholePlaneCameraModel, "FishEye62": Fisheye62CameraModel}[js["DistortionModel"]]
    coeffs = [js[name] for name in cls.distortion_model._fields]
    return cls(js["ImageSizeX"], js["ImageSizeY"], (js["fx"], js["fy"]), (js["cx"], js["cy"]), coeffs)
