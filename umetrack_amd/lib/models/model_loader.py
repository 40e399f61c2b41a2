"""load_pretrained_model with the reference's signature (lib/models/model_loader.py:53-88)."""
import torch

from ... import arch
from .model_opts import ModelOpts  # noqa: F401  (re-exported like the reference module)
from .umetrack_model import UmeTrackModel


def load_pretrained_model(model_path: str) -> UmeTrackModel:
    """Read a plain state_dict checkpoint and build the native model (strict key check).

    The reference calls ``torch.load(fp)``; only tensors are expected in the
    file, so the safe ``weights_only=True`` loader is used here.
    """
    with open(model_path, "rb") as fp:
        state = torch.load(fp, map_location="cpu", weights_only=True)
    return UmeTrackModel(state)


def model_from_state_dict(state_dict) -> UmeTrackModel:
    """Build a model from an in-memory state_dict (tensors or numpy arrays)."""
    return UmeTrackModel(state_dict)


def synthetic_model(seed: int = 0) -> UmeTrackModel:
    """Random-weight model for benchmarks and tests (pretrained_weights.torch is not distributed)."""
    return UmeTrackModel(arch.make_synthetic_state(seed))
