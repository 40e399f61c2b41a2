"""The parameter table must be exactly the reference's state_dict (row A-W)."""
import json
import os

import numpy as np

from umetrack_amd import arch
from conftest import GOLDEN


def test_state_spec_matches_reference_state_dict():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    spec = arch.state_spec()
    assert list(spec.keys()) == list(ref.keys())
    for k, (shape, dtype) in spec.items():
        assert list(shape) == ref[k][0], k
        assert dtype == ref[k][1], k
    n_float = sum(int(np.prod(s)) for s, d in spec.values() if d == "float32")
    assert len(spec) == 252
    assert sum(int(np.prod(s)) if s else 1 for s, _ in spec.values()) == 4259410
    assert n_float == 4259410 - 39                    # 39 BN layers each hold one int64 counter


def test_synthetic_state_is_deterministic():
    a, b = arch.make_synthetic_state(3), arch.make_synthetic_state(3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = arch.make_synthetic_state(4)
    assert any(not np.array_equal(a[k], c[k]) for k in a)
    assert arch.regressor_out_dims(False) == 62 and arch.regressor_out_dims(True) == 63
    assert arch.regressor_slices(True)["landmark_uncertainty_sigmas"] == (42, 63)
