"""GPU crop warp: the batched counterpart of _warp_image (lib/tracker/tracker.py:54-80).

``warp_crops`` resamples many fisheye source views into pinhole crops in one
launch of csrc/warp.hip.  Cameras are passed as packed float64 rows so that the
coordinate map is evaluated from the same double-precision parameters the
reference's numpy code uses.
"""
from typing import Sequence

import numpy as np
import torch

from ... import _native as nat

SRC_CAM_WORDS = 30   # w,h,fx,fy,cx,cy,k1,k2,k3,k4,p1,p2,k5,k6, cam_to_world(16)
DST_CAM_WORDS = 20   # fx,fy,cx,cy, cam_to_world(16)
MODE_CV2 = nat.WARP_CV2_FIXED_POINT
MODE_FLOAT = nat.WARP_FLOAT
MODE_CV2_F64 = nat.WARP_CV2_FIXED_POINT_F64      # every map pixel in float64 (MODE_CV2 is bit-identical, faster)
NO_DEPTH_CHECK = nat.WARP_NO_DEPTH_CHECK         # OR into the mode: _warp_image(depth_check=False), tracker.py:59,71-73


def pack_fisheye(width, height, f, c, coeffs: Sequence[float], cam_to_world) -> np.ndarray:
    """coeffs in Fisheye62 field order k1,k2,k3,k4,p1,p2,k5,k6 (camera.py:111-118)."""
    f = np.broadcast_to(np.asarray(f, np.float64), 2)
    row = np.empty(SRC_CAM_WORDS, np.float64)
    row[:6] = (width, height, f[0], f[1], c[0], c[1])
    row[6:14] = np.asarray(coeffs, np.float64)
    row[14:] = np.asarray(cam_to_world, np.float64).reshape(-1)[:16]
    return row


def pack_pinhole(f, c, cam_to_world) -> np.ndarray:
    f = np.broadcast_to(np.asarray(f, np.float64), 2)
    row = np.empty(DST_CAM_WORDS, np.float64)
    row[:4] = (f[0], f[1], c[0], c[1])
    row[4:] = np.asarray(cam_to_world, np.float64).reshape(-1)[:16]
    return row


def warp_crops(src_images: torch.Tensor, src_offset, src_pitch, src_cams, dst_cams,
               out_hw=(96, 96), mode: int = MODE_CV2, out: torch.Tensor = None) -> torch.Tensor:
    """src_images: uint8 device buffer holding every source view; crop k reads the image that
    starts ``src_offset[k]`` bytes into it with row pitch ``src_pitch[k]``.
    src_cams [n,30] / dst_cams [n,20] float64.  Returns float32 [n,H,W] in [0,1]."""
    dev = nat.require_gpu(src_images.device if src_images.is_cuda else None)
    img = nat.dev_tensor(src_images, torch.uint8, dev)
    off = nat.dev_tensor(src_offset, torch.int64, dev).reshape(-1)
    pit = nat.dev_tensor(src_pitch, torch.int32, dev).reshape(-1)
    sc = nat.dev_tensor(src_cams, torch.float64, dev).reshape(-1, SRC_CAM_WORDS)
    dc = nat.dev_tensor(dst_cams, torch.float64, dev).reshape(-1, DST_CAM_WORDS)
    n = sc.shape[0]
    assert dc.shape[0] == n and off.numel() == n and pit.numel() == n
    if out is None:
        out = torch.empty(n, out_hw[0], out_hw[1], dtype=torch.float32, device=dev)
    else:
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (n, out_hw[0], out_hw[1])
    with torch.cuda.device(dev):
        nat.check(nat.lib().umt_warp_crops(nat.ptr(img), nat.ptr(off), nat.ptr(pit), nat.ptr(sc), nat.ptr(dc), n,
                                           out_hw[0], out_hw[1], mode, nat.ptr(out), nat.stream_ptr(dev)),
                  "umt_warp_crops")
    return out
