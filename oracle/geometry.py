"""CPU restatement of the image-resampling stage (SURVEY.md section 8, rows A1, A1').

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy, fp64 coordinate math like
the reference's numpy code.

Pinned:
* ``fisheye_warp_maps`` against tests/golden/warp_maps.npz (the reference's
  importable camera classes, lib/common/camera.py, driven as tracker.py:61-78
  does).
* ``resample_pinhole`` against tests/golden/resample.npz
  (lib/batched_dataset/data_transform.py:78-144).

PARITY UNPINNED: ``remap_linear_u8`` restates the published behaviour of
OpenCV's ``cv2.remap(..., INTER_LINEAR)`` for 8-bit images (opencv-python is an
unpinned pip dependency, README.md:10; call site tracker.py:80).  cv2 is not in
this image and the reference holds no output of it, so this function is checked
by properties only (integer maps reproduce the source; constant images stay
constant; agreement with float bilinear within one grey level).
"""
from dataclasses import dataclass
from typing import Sequence, Tuple

import numpy as np


@dataclass
class Fisheye62:
    """camera.py:106-141, 330-331.  coeffs = (k1,k2,k3,k4,p1,p2,k5,k6)."""
    width: int
    height: int
    f: Tuple[float, float]
    c: Tuple[float, float]
    coeffs: Sequence[float]
    cam_to_world: np.ndarray


@dataclass
class Pinhole:
    """camera.py:320-327."""
    width: int
    height: int
    f: Tuple[float, float]
    c: Tuple[float, float]
    cam_to_world: np.ndarray


def fisheye_project(cam: Fisheye62, pts_eye: np.ndarray) -> np.ndarray:
    """eye -> window for the arctan + 6/2 distortion model (camera.py:78-85, 120-141, 269-273)."""
    x, y, z = pts_eye[..., 0], pts_eye[..., 1], pts_eye[..., 2]
    r = np.sqrt(x * x + y * y)
    s = np.arctan2(r, z) / np.maximum(r, 2.0 ** -128)
    px, py = x * s, y * s
    k1, k2, k3, k4, p1, p2, k5, k6 = cam.coeffs
    r2 = np.clip(px * px + py * py, -np.pi ** 2, np.pi ** 2)
    r4 = r2 * r2
    r6 = r2 * r4
    r8 = r4 * r4
    r10 = r4 * r6
    r12 = r6 * r6
    radial = 1 + k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8 + k5 * r10 + k6 * r12
    ux, uy = px * radial, py * radial
    x2, y2, xy = ux * ux, uy * uy, ux * uy
    rr = x2 + y2
    qx = ux + 2 * p2 * xy + p1 * (rr + 2 * x2)
    qy = uy + 2 * p1 * xy + p2 * (rr + 2 * y2)
    return np.stack((qx * cam.f[0] + cam.c[0], qy * cam.f[1] + cam.c[1]), -1)


def fisheye_warp_maps(src: Fisheye62, dst: Pinhole, depth_check: bool = True):
    """map_x, map_y (H,W) float32 exactly as tracker.py:61-78 builds them."""
    W, H = dst.width, dst.height
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    win = np.column_stack((px.ravel(), py.ravel())).astype(np.float64)
    q = (win - np.asarray(dst.c)) / np.asarray(dst.f)                 # camera.py:275-281
    v = np.concatenate((q, np.ones((q.shape[0], 1))), 1)
    v = v / np.maximum(5.43e-20, np.sqrt((v * v).sum(-1, keepdims=True)))   # affine.py:22-24
    world = v @ dst.cam_to_world[:3, :3].T + dst.cam_to_world[:3, 3]  # camera.py:263-267
    # camera.py:257-261: transform_vec3(xf.T, v - t) == (v - t) @ R
    eye = (world - src.cam_to_world[:3, 3]) @ src.cam_to_world[:3, :3]
    w = fisheye_project(src, eye)
    if depth_check:
        w[eye[:, 2] < 0] = -1
    w = w.astype(np.float32)
    return w[:, 0].reshape(H, W), w[:, 1].reshape(H, W)


def remap_linear_u8(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """[external, unpinned] OpenCV 8-bit INTER_LINEAR remap, constant-0 border.

    Coordinates are rounded (half to even) to 1/32 pixel; the four weights are
    the exact products of the 5-bit fractions scaled to 2**15; the weighted sum
    is rounded with +2**14 and shifted by 15.
    """
    assert src.dtype == np.uint8 and src.ndim == 2
    h, w = src.shape
    sx = np.rint(map_x.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(map_y.astype(np.float32) * np.float32(32)).astype(np.int64)
    sx = np.clip(sx, -(1 << 20), 1 << 20)
    sy = np.clip(sy, -(1 << 20), 1 << 20)
    ix, iy = sx >> 5, sy >> 5
    fx, fy = sx & 31, sy & 31

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0).astype(np.int64)

    acc = (tap(iy, ix) * ((32 - fy) * (32 - fx)) + tap(iy, ix + 1) * ((32 - fy) * fx)
           + tap(iy + 1, ix) * (fy * (32 - fx)) + tap(iy + 1, ix + 1) * (fy * fx)) * 32
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def remap_linear_f32(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """Plain float bilinear with constant-0 border (the 'float' warp mode)."""
    h, w = src.shape
    mx, my = map_x.astype(np.float64), map_y.astype(np.float64)
    x0, y0 = np.floor(mx), np.floor(my)
    ax, ay = mx - x0, my - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    s = src.astype(np.float64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok, s[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0.0)

    out = (tap(y0, x0) * (1 - ax) * (1 - ay) + tap(y0, x0 + 1) * ax * (1 - ay)
           + tap(y0 + 1, x0) * (1 - ax) * ay + tap(y0 + 1, x0 + 1) * ax * ay)
    return out.astype(np.float32)


def warp_crop(src_cam: Fisheye62, dst_cam: Pinhole, image_u8: np.ndarray, mode: str = "cv2") -> np.ndarray:
    """tracker.py:54-80 + the /255 at tracker.py:211 -> float32 (H,W) in [0,1]."""
    mx, my = fisheye_warp_maps(src_cam, dst_cam)
    if mode == "cv2":
        return remap_linear_u8(image_u8, mx, my).astype(np.float32) / 255.0
    return remap_linear_f32(image_u8, mx, my) / np.float32(255.0)


def resample_matrix(k_orig: np.ndarray, world_to_eye_orig: np.ndarray,
                    k_new: np.ndarray, eye_to_world_new: np.ndarray) -> np.ndarray:
    """data_transform.py:57-75: K_o . E_o . E_n^-1 . K_n^-1 as float32 4x4."""
    kin = np.eye(4)
    kin[:3, :3] = np.linalg.inv(k_new)
    ko = np.eye(4)
    ko[:3, :3] = k_orig
    return (ko @ world_to_eye_orig @ eye_to_world_new @ kin).astype(np.float32)


def resample_pinhole(images: np.ndarray, xfs: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """data_transform.py:78-144, stated per pixel instead of by fancy indexing.

    images (n,h,w) float32, xfs (n,4,4) float32 -> (n,H,W) float32.  Coordinates
    are float64 because the reference's tensordot of a float32 matrix with an
    int32 grid promotes to float64.
    """
    n, h, w = images.shape
    H, W = out_hw
    r = xfs[:, :3, :3].astype(np.float64)
    t = xfs[:, :3, 3].astype(np.float64)
    v, u = np.mgrid[0:H, 0:W]
    grid = np.stack((u, v, np.ones_like(u)), -1).astype(np.float64)          # (H,W,3) = (u,v,1)
    p = np.einsum("nij,hwj->nhwi", r, grid) + t[:, None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = p[..., 0] / p[..., 2]
        y = p[..., 1] / p[..., 2]
    ok = (x >= 0) & (x < w - 1) & (y >= 0) & (y < h - 1)
    xs, ys = np.where(ok, x, 0.0), np.where(ok, y, 0.0)
    x0, y0 = xs.astype(np.int32), ys.astype(np.int32)
    x1, y1 = x0 + 1, y0 + 1
    ni = np.arange(n)[:, None, None]
    img = images
    val = (img[ni, y0, x0] * (x1 - xs) * (y1 - ys) + img[ni, y0, np.minimum(x1, w - 1)] * (xs - x0) * (y1 - ys)
           + img[ni, np.minimum(y1, h - 1), x0] * (x1 - xs) * (ys - y0)
           + img[ni, np.minimum(y1, h - 1), np.minimum(x1, w - 1)] * (xs - x0) * (ys - y0)) \
        / ((x1 - x0) * (y1 - y0))
    return np.where(ok, val, 0.0).astype(np.float32)
