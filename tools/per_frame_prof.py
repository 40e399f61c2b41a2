#!/usr/bin/env python3
"""Where a per-frame tracker call spends its time: cProfile of the host side (default) or a plain run for
rocprofv3 --kernel-trace --stats (argument `plain`).   python3 tools/per_frame_prof.py [plain] [frames]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

plain = len(sys.argv) > 1 and sys.argv[1] == "plain"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pipe = bench.Pipeline(256, 64, 256, seed=0, device=dev, precision=os.environ.get("UMT_PF_PREC", "fp32"))
pipe.step()
torch.cuda.synchronize()
print(bench.per_frame_row(pipe, frames)["ms_per_frame"], "ms per frame (warm-up pass)")
if plain:
    print(bench.per_frame_row(pipe, frames)["ms_per_frame"], "ms per frame")
else:
    pr = cProfile.Profile()
    pr.enable()
    row = bench.per_frame_row(pipe, frames)
    pr.disable()
    print(row["ms_per_frame"], "ms per frame under cProfile")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
