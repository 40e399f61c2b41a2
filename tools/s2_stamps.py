#!/usr/bin/env python3
"""Phase stamps of conv3x3s2_walk_bf16 (library built with -DUMT_S2_STAMP=<cin>) or conv3x3s2p_ds_f32 (-DUMT_S2P_STAMP=<cin>);
experiments only:  python tools/s2_stamps.py tools/bin/libumt_s2_STAMP64.so [bf16|fp32]"""
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
gm.precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
img = torch.rand(4096, 96, 96, device="cuda")
for _ in range(3):
    gm.backbone(img)
torch.cuda.synchronize()
L = nat.lib()
st = (ctypes.c_ulonglong * 64)()
L.umt_s2_stamps.argtypes = [ctypes.c_void_p]
L.umt_s2_stamps(st)
print("second tile of workgroup 5, wave 0 (s_memtime ticks):")
names = {0: "tile start"}
for c in range(4):
    names.update({1 + 4 * c: f"chunk {c}: rows written", 2 + 4 * c: f"chunk {c}: barrier", 3 + 4 * c: f"chunk {c}: load issue + 9 taps",
                  4 + 4 * c: f"chunk {c}: barrier"})
names.update({17: "next tile rows requested", 18: "epilogue conv1", 20: "epilogue(s)", 21: "barrier"})
prev = st[0]
for k in sorted(names):
    if st[k]:
        print(f"  {names[k]:36s} +{st[k] - prev:7d}   (at {st[k] - st[0]:7d})")
        prev = st[k]
tiles = [st[32 + i] for i in range(32) if st[32 + i]]
print("  tile starts, differences:", [tiles[i + 1] - tiles[i] for i in range(len(tiles) - 1)])
