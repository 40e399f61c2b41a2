#!/usr/bin/env python3
"""Per-stage kernel timings against their rooflines at the bench sizes (HIP events on the launch stream).
   python tools/stage_roofline.py [hands]      -> one JSON line per stage"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from umetrack_amd.lib.common import metric_utils as mu  # noqa: E402
from umetrack_amd.lib.common.hand_skinning import skin_landmarks  # noqa: E402
from umetrack_amd.lib.tracker import warp as gpu_warp  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    frames = int(sys.argv[1]) // 2 if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda", 0)
    pipe = bench.Pipeline(frames, 64, min(frames, 2048), seed=0, device=dev)
    S, cs = pipe.S, pipe.chunk
    rows = []
    # warp: 73,728 B per crop (fp32 out + 4 one-byte taps per pixel)
    for mode, name in ((gpu_warp.MODE_CV2, "warp cv2-fixed-point"), (gpu_warp.MODE_FLOAT, "warp float")):
        t = timed(lambda: gpu_warp.warp_crops(pipe.images, pipe.src_off[:2 * cs], pipe.src_pitch[:2 * cs],
                                              pipe.src_cams[:2 * cs], pipe.dst_cams[:2 * cs], mode=mode, out=pipe.crops))
        rows.append({"stage": name, "units": 2 * cs, "unit": "crop", "bytes_per_unit": 73728, "seconds": t})
    # FK: 404 B per hand with a shared hand model (88 + 64 in, 252 out)
    out = torch.empty(S, 21, 3, device=dev)
    t = timed(lambda: skin_landmarks(pipe.hm, pipe.gt_angles, pipe.gt_wrist, hand_idx=pipe.hand_idx, mirror_right=True, out=out))
    rows.append({"stage": "fk shared model", "units": S, "unit": "hand", "bytes_per_unit": 404, "seconds": t})
    # metrics: 2 x 252 B landmarks + 1 B valid per hand-frame
    pipe.step()
    t = timed(lambda: mu.metric_partials(pipe.gt_lm, pipe.pred_lm, pipe.valid))
    rows.append({"stage": "metric partials", "units": S, "unit": "hand-frame", "bytes_per_unit": 505, "seconds": t})
    # stem: 36,864 B in + 294,912 B out per crop
    from umetrack_amd import _native as nat
    for r in rows:
        gbs = r["units"] * r["bytes_per_unit"] / r["seconds"] / 1e9
        r.update(achieved_GBs=gbs, hbm_peak_GBs=HBM_PEAK_GBS, frac=gbs / HBM_PEAK_GBS, us=r["seconds"] * 1e6)
        print(json.dumps(r))


if __name__ == "__main__":
    main()
