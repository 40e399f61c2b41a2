"""CPU-side checks of the C-ABI library: it loads, exports what the header declares,
enforces the strict key table, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from umetrack_amd import _native as nat
from umetrack_amd import arch


def _header_functions():
    text = open(os.path.join(ROOT, "include", "umetrack_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(umt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 20
    handle = ctypes.CDLL(nat.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/umetrack_hip.h but not exported"
    assert sorted(nat.SIGNATURES) == names, "ctypes table and header disagree"
    assert nat.lib().umt_abi_version() == nat.ABI_VERSION == 4


def test_native_key_table_equals_python_spec():
    L = nat.lib()
    h = ctypes.c_void_p()
    nat.check(L.umt_model_create(ctypes.byref(h)))
    cnt = ctypes.c_int()
    nat.check(L.umt_model_param_count(h, ctypes.byref(cnt)))
    spec = arch.state_spec()
    assert cnt.value == len(spec) == 252
    for i, (key, (shape, dtype)) in enumerate(spec.items()):
        k, nd, shp, isint = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)(), ctypes.c_int()
        nat.check(L.umt_model_param_info(h, i, ctypes.byref(k), ctypes.byref(nd), ctypes.byref(shp), ctypes.byref(isint)))
        assert k.value.decode() == key
        assert tuple(shp[:nd.value]) == tuple(shape)
        assert bool(isint.value) == (dtype == "int64")
    L.umt_model_destroy(h)


def test_strict_state_dict_errors():
    from umetrack_amd.lib.models.umetrack_model import UmeTrackModel
    sd = arch.make_synthetic_state(0)
    missing = dict(sd)
    missing.pop("_temporal._temporal_module.2.bias")
    with pytest.raises(KeyError, match="Missing"):
        UmeTrackModel(missing)
    extra = dict(sd)
    extra["bogus.weight"] = np.zeros(3, np.float32)
    with pytest.raises(KeyError, match="Unexpected"):
        UmeTrackModel(extra)
    bad = dict(sd)
    bad["_feature_extractor._image_backbone.1.bias"] = np.zeros(71, np.float32)
    with pytest.raises(KeyError, match="size mismatch"):
        UmeTrackModel(bad)
    # the native table rejects unknown keys on its own as well
    L = nat.lib()
    h = ctypes.c_void_p()
    nat.check(L.umt_model_create(ctypes.byref(h)))
    buf = np.zeros(4, np.float32)
    shp = (ctypes.c_int64 * 1)(4)
    rc = L.umt_model_set_param(h, b"nope", buf.ctypes.data_as(ctypes.c_void_p), 1, shp, 0)
    assert rc == 3 and b"Unexpected key" in L.umt_last_error()
    assert L.umt_model_finalize(h, 0) == 3 and b"Missing key" in L.umt_last_error()
    L.umt_model_destroy(h)


def test_workspace_sizes_are_monotonic():
    L = nat.lib()
    a, b = L.umt_forward_workspace_bytes(4, 2), L.umt_forward_workspace_bytes(64, 32)
    assert 0 < a < b
    assert L.umt_forward_workspace_bytes(4, 2) == L.umt_backbone_workspace_bytes(4) + L.umt_head_workspace_bytes(4, 2)
    assert L.umt_metric_workspace_bytes(3, 100) >= 108 * 8


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from umetrack_amd.lib.models.model_loader import synthetic_model
    from umetrack_amd.lib.common.hand_skinning import skin_landmarks
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    m = synthetic_model(0)
    with pytest.raises(nat.UmtError, match="no MI355X"):
        m.to("cuda")
    with pytest.raises(nat.UmtError):
        skin_landmarks(load_generic_hand_model(), torch.zeros(22), torch.eye(4))
