#!/usr/bin/env python3
"""Times umt_warp_crops of several builds of the library on the bench workload's cameras (experiments only).
  python tools/warp_bench.py tools/bin/libumt_warp_BASE.so tools/bin/libumt_warp_NOMAP.so ..."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from umetrack_amd import _native as nat  # noqa: E402

frames = int(os.environ.get("FRAMES", "4096"))
dev = torch.device("cuda", 0)
pipe = bench.Pipeline(frames, 64, frames, seed=0, device=dev, host_crop_cameras=True)
n = 2 * pipe.S
P = ctypes.c_void_p
out = torch.empty(n, 96, 96, device=dev)
ref = None
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(ROOT, path))
    L.umt_warp_crops.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]
    for mode in (0, 2, 1):
        def run():
            rc = L.umt_warp_crops(nat.ptr(pipe.images), nat.ptr(pipe.src_off), nat.ptr(pipe.src_pitch), nat.ptr(pipe.src_cams),
                                  nat.ptr(pipe.dst_cams), n, 96, 96, mode, nat.ptr(out), nat.stream_ptr(dev))
            assert rc == 0
        for _ in range(60 if ref is None and mode == 0 else 10):   # clocks ramp up from idle over the first ~100 ms
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        note = ""
        if mode == 0:
            if ref is None:
                ref = out.clone()
            else:
                note = f"  differs from first lib in {(out != ref).float().mean().item():.4f} of pixels"
        if hasattr(L, "umt_warp_stamps") and mode == 0:
            st = (ctypes.c_ulonglong * 16)()
            L.umt_warp_stamps(st)
            names = ["nodes f64", "errors+cells", "interpolate", "queue", "bbox+reduce", "tile copy", "gather", "store"]
            print("   s_memtime ticks (100 MHz) of workgroup 1000:", ", ".join(f"{names[i]} {st[i + 1] - st[i]}" for i in range(8)))
        print(f"{os.path.basename(path):28s} mode {mode}: {ms:7.3f} ms per {n} crops = {n * 73728 / ms / 1e9:6.2f} TB/s{note}", flush=True)
