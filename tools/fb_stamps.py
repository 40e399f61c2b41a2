#!/usr/bin/env python3
"""Phase stamps of block32_fused_bf16 (diagnostic library built with -DUMT_FB_STAMP; experiments only):
  python tools/fb_stamps.py tools/bin/libumt_fb_STAMP.so"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umetrack_amd import _native as nat, arch  # noqa: E402

nat.LIB_PATH = os.path.join(ROOT, sys.argv[1])          # before the first nat.lib() call
from umetrack_amd.lib.models.model_loader import model_from_state_dict  # noqa: E402

gm = model_from_state_dict(arch.make_synthetic_state(0)).to("cuda")
gm.precision = "bf16"
img = torch.rand(4096, 96, 96, device="cuda")
for _ in range(3):
    gm.backbone(img)
torch.cuda.synchronize()
L = nat.lib()
st = (ctypes.c_ulonglong * 16)()
L.umt_fb_stamps.argtypes = [ctypes.c_void_p]
L.umt_fb_stamps(st)
names = ["prefetch issue + constants + row table", "conv1 MFMAs", "conv1 epilogue", "barrier", "conv2 setup", "conv2 MFMAs",
         "conv2 epilogue + stores", "barrier", "write next input + barrier"]
print("block32_fused_bf16, one tile of workgroup 3 (shader cycles):")
for i, n in enumerate(names):
    print(f"  {n:42s} {st[i + 1] - st[i]:7d}")
print(f"  {'tile total':42s} {st[9] - st[0]:7d}")
