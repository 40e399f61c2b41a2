"""Batched pinhole->pinhole resampling on the GPU (torch_data path, row A1').

``resample_images_batched`` has the argument meaning of
lib/batched_dataset/data_transform.py:78-144 but returns the new images instead
of filling a numpy array in place.
"""
from typing import Tuple

import torch

from ... import _native as nat


def resample_images_batched(images_orig: torch.Tensor, resample_xfs: torch.Tensor,
                            crop_size: Tuple[int, int] = (96, 96)) -> torch.Tensor:
    """images_orig [n,h,w] float32, resample_xfs [n,4,4] float32 -> [n,H,W] float32 (not /255)."""
    dev = nat.require_gpu(images_orig.device if images_orig.is_cuda else None)
    img = nat.dev_tensor(images_orig, torch.float32, dev)
    xf = nat.dev_tensor(resample_xfs, torch.float32, dev)
    n, h, w = img.shape
    assert tuple(xf.shape) == (n, 4, 4)
    out = torch.empty(n, crop_size[0], crop_size[1], dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().umt_resample_pinhole(nat.ptr(img), nat.ptr(xf), n, h, w, crop_size[0], crop_size[1],
                                                 nat.ptr(out), nat.stream_ptr(dev)), "umt_resample_pinhole")
    return out
