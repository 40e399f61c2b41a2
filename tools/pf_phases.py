#!/usr/bin/env python3
"""Host-side phases of HandTracker.track_frame, wall-clocked (no profiler): where the 0.66 ms of a frame go.
   python3 tools/pf_phases.py [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from umetrack_amd.lib.tracker import tracker as T  # noqa: E402

acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrap


T.HandTracker._make_inputs = timed("make_inputs (frame copy, parameters, uploads, warp launch)", T.HandTracker._make_inputs)
T.HandTracker._finish = timed("finish (FK launch, status, download, WAIT, result dicts)", T.HandTracker._finish)
T._FrameIO.download = timed("  of which download + stream synchronise", T._FrameIO.download)
T.np.copyto = timed("  of which frame copy into the pinned block", T.np.copyto)
T._FrameIO.upload = timed("  of which the two upload calls", T._FrameIO.upload)
T.gpu_warp.warp_crops = timed("  of which warp_crops (wrapper + launch)", T.gpu_warp.warp_crops)
T._common_parent = timed("  of which _common_parent", T._common_parent)
T._FrameIO.plan = timed("  of which plan", T._FrameIO.plan)
T.HandTracker._device_hand_model = timed("  _device_hand_model", T.HandTracker._device_hand_model)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pipe = bench.Pipeline(256, 64, 256, seed=0, device=dev)
pipe.step()
torch.cuda.synchronize()
bench.per_frame_row(pipe, frames)
acc.clear()
row = bench.per_frame_row(pipe, frames)
n = frames + min(8, frames)
print(f"{row['ms_per_frame']:.3f} ms per frame")
for k, v in acc.items():
    print(f"  {k:70s} {v / n * 1e6:7.1f} us per frame")
