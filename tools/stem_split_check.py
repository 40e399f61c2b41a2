#!/usr/bin/env python3
"""bf16 backbone features with the split-operand stem against the fp32-chain stem (UMT_STEM_NOSPLIT) and the fp32 mode:
how far the split moves the features compared with what the bf16 mode itself costs.  GPU box; experiments only."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umetrack_amd import arch  # noqa: E402
from umetrack_amd.lib.models.model_loader import model_from_state_dict  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "child":
    gm = model_from_state_dict(arch.make_synthetic_state(4)).to("cuda")
    img = torch.rand(512, 96, 96, generator=torch.Generator().manual_seed(78))
    gm.precision = "bf16"
    np.save(sys.argv[2], gm.backbone(img.cuda()).cpu().numpy())
    sys.exit(0)
out = "/tmp/stem_nosplit.npy"
subprocess.run([sys.executable, __file__, "child", out], check=True, env=dict(os.environ, UMT_STEM_NOSPLIT="1"), timeout=200)
gm = model_from_state_dict(arch.make_synthetic_state(4)).to("cuda")
img = torch.rand(512, 96, 96, generator=torch.Generator().manual_seed(78))
f32 = gm.backbone(img.cuda()).cpu().numpy()
gm.precision = "bf16"
split = gm.backbone(img.cuda()).cpu().numpy()
chain = np.load(out)
mag = np.abs(f32).max()
print(f"features max |x| {mag:.4f}")
print(f"bf16(chain stem) vs fp32: max {np.abs(chain - f32).max() / mag:.3e} mean {np.abs(chain - f32).mean() / mag:.3e} of max |x|")
print(f"bf16(split stem) vs fp32: max {np.abs(split - f32).max() / mag:.3e} mean {np.abs(split - f32).mean() / mag:.3e}")
print(f"split vs chain stem:      max {np.abs(split - chain).max() / mag:.3e} mean {np.abs(split - chain).mean() / mag:.3e}; "
      f"{(split != chain).mean():.4f} of the features differ")
