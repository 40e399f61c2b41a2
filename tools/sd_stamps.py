#!/usr/bin/env python3
"""Role stamps of stage1_depth_bf16 (diagnostic library built with -DUMT_SD_STAMP; experiments only):
  python tools/sd_stamps.py tools/bin/libumt_sd_STAMP.so"""
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
st = (ctypes.c_ulonglong * 32)()
L.umt_sd_stamps.argtypes = [ctypes.c_void_p]
L.umt_sd_stamps(st)
roles = ["c1a", "c2a", "c1b", "c2b", "s2 a", "loader", "s2 b", "ds"]
t0 = min(st[4 * w] for w in range(8) if st[4 * w])
print("stage1_depth_bf16, step 100 of workgroup 3 (s_memtime ticks, 100 MHz -> x ~20 shader cycles):")
for w, r in enumerate(roles):
    a = [st[4 * w + i] - t0 for i in range(4)]
    print(f"  {r:7s} start {a[0]:6d}  after phase 1 {a[1]:6d}  after phase 2 {a[2]:6d}  at barrier {a[3]:6d}")
