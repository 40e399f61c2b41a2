#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 passes: a few whole-model forwards on random crops.
   rocprofv3 --kernel-trace --stats -- python3 tools/profile_forward.py [crops] [iters] [fp32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umetrack_amd import arch  # noqa: E402
from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model  # noqa: E402
from umetrack_amd.lib.models.model_loader import model_from_state_dict  # noqa: E402
from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
m = model_from_state_dict(arch.make_synthetic_state(0)).to(dev)
m.precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
bs = n // 2
g = torch.Generator(device=dev).manual_seed(0)
fd = InputFrameData(torch.rand(n, 96, 96, device=dev, generator=g),
                    torch.tensor([[250.0, 0, 47.5], [0, 250.0, 47.5], [0, 0, 1]], device=dev).repeat(n, 1, 1),
                    torch.eye(4, device=dev).repeat(n, 1, 1))
desc = InputFrameDesc(torch.arange(bs, device=dev)[:, None] * 2 + torch.tensor([0, 2], device=dev),
                      torch.arange(bs, device=dev), torch.zeros(bs, dtype=torch.bool, device=dev),
                      torch.arange(bs, device=dev) % 2)
hm = scaled_hand_model(load_generic_hand_model(), 0.001)
skel = InputSkeletonData(hm.joint_rotation_axes.to(dev), hm.joint_rest_positions.to(dev))
m.timing_enable(True)
for _ in range(iters):
    out = m.regress_pose_use_skeleton(fd, desc, skel, required_rows=bs)
torch.cuda.synchronize()
b, f, c = m.timing_read()
print(f"crops {n} iters {c} backbone {b / c:.3f} ms forward {f / c:.3f} ms -> {969.23e6 * n / (b / c * 1e-3) / 1e12:.1f} TFLOP/s")
