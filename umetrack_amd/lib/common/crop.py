"""Crop-camera construction on the host (row A0); follows lib/common/crop.py:15-82."""
from typing import Tuple

import numpy as np

from . import affine
from . import camera


def gen_intrinsics_from_bounding_pts(pts_eye: np.ndarray, image_w: int, image_h: int, min_focal: float = 5):
    """Focal length that just keeps every point inside the image, principal point at the centre."""
    ndc = pts_eye[..., :2] / pts_eye[..., 2:]
    cx_cy = (np.array([image_w, image_h], dtype=pts_eye.dtype) - 1) / 2
    fx_fy = cx_cy / np.abs(ndc).max()
    if np.any(pts_eye[..., 2:] < 0.0001) or np.any(fx_fy < min_focal):
        raise ValueError("Unable to create crop camera", fx_fy)
    return fx_fy, cx_cy


def gen_crop_parameters_from_points(camera_orig: camera.CameraModel, pts_world, new_image_size: Tuple[int, int],
                                    mirror_img_x: bool, camera_angle: float = 0,
                                    focal_multiplier: float = 0.95) -> camera.PinholePlaneCameraModel:
    """Pinhole camera at camera_orig's position that looks at the bounding-box centre of
    pts_world, optionally mirrored in x (right hands), zoomed out by focal_multiplier."""
    world_to_eye = np.linalg.inv(camera_orig.camera_to_world_xf)
    centre = (pts_world.min(axis=0) + pts_world.max(axis=0)) / 2.0
    new_w2e = affine.make_look_at_matrix(world_to_eye, centre, camera_angle)
    if mirror_img_x:
        flip = np.eye(4, dtype=np.float32)
        flip[0, 0] = -1
        new_w2e = flip @ new_w2e
    fx_fy, cx_cy = gen_intrinsics_from_bounding_pts(affine.transform3(new_w2e, pts_world),
                                                    new_image_size[0], new_image_size[1])
    return camera.PinholePlaneCameraModel(width=new_image_size[0], height=new_image_size[1],
                                          f=focal_multiplier * fx_fy, c=cx_cy, distort_coeffs=[],
                                          camera_to_world_xf=np.linalg.inv(new_w2e))
