"""RegressorOutput, same fields as lib/models/regressor.py:124-129."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class RegressorOutput:
    joint_angles: torch.Tensor
    wrist_xfs: torch.Tensor
    skel_scales: Optional[torch.Tensor] = None
    landmark_uncertainty_sigmas: Optional[torch.Tensor] = None
