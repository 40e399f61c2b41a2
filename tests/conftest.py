import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def model_sequence():
    """tests/golden/model_sequence.npz as a list of per-step dicts."""
    flat = load_npz("model_sequence.npz")
    steps = []
    for i in range(int(flat["n_steps"])):
        pre = f"{i}/"
        steps.append({k[len(pre):]: v for k, v in flat.items() if k.startswith(pre)})
    return steps, int(flat["seed"])


@pytest.fixture(scope="session")
def golden_sequence():
    return model_sequence()


@pytest.fixture(scope="session")
def hand_model_np():
    import json
    with open(os.path.join(ROOT, "umetrack_amd", "data", "generic_hand_model_small.json")) as fh:
        d = json.load(fh)
    return {k: np.asarray(v) for k, v in d.items()}
