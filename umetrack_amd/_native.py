"""ctypes binding of libumetrack_hip.so (the C ABI declared in include/umetrack_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible
when a kernel is requested, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (UMT_LIB_PATH: diagnostics only - A/B runs of an experiment build of the same ABI, e.g. tools/bin/*.so)
LIB_PATH = os.environ.get("UMT_LIB_PATH") or os.path.join(_HERE, "libumetrack_hip.so")

PREC_FP32, PREC_BF16, PREC_FP32_BF16X3 = 0, 1, 2
HEAD_KNOWN_SKELETON, HEAD_PREDICT_SCALE = 0, 1
WARP_CV2_FIXED_POINT, WARP_FLOAT, WARP_CV2_FIXED_POINT_F64 = 0, 1, 2
WARP_NO_DEPTH_CHECK = 0x100
METRIC_WORDS = 108
ABI_VERSION = 4          # UMT_ABI_VERSION of include/umetrack_hip.h these signatures were written for
STATUS_BITS = {1: "use_memory on a memory row whose previous extrinsics are singular (torch.inverse raises in the "
                  "reference, temporal.py:66); the sample ran as use_memory=False",
               2: "memory_idx outside the reserved memory rows",
               4: "sample_range row outside [0, n] or not 1 or 2 views",
               8: "a regressor output row was not finite (NaN/inf in the crop, the weights or the state)",
               16: "a landmark bone index outside [0, 16]"}

STATUS_NONFINITE_OUTPUT = 8
# the conditions under which the reference itself raises (torch.inverse on a singular matrix, index errors); a
# non-finite regressor row is returned as NaN there and the run goes on
STATUS_RAISING = 1 | 2 | 4 | 16

_ERR_NAMES = {1: "invalid argument", 2: "bad state", 3: "state_dict key", 4: "HIP runtime",
              5: "workspace", 6: "unsupported"}

# name -> (restype, argtypes); kept in the order of include/umetrack_hip.h
_P = c_void_p
SIGNATURES = {
    "umt_last_error": (c_char_p, []),
    "umt_abi_version": (c_int, []),
    "umt_model_create": (c_int, [POINTER(c_void_p)]),
    "umt_model_destroy": (c_int, [_P]),
    "umt_model_param_count": (c_int, [_P, POINTER(c_int)]),
    "umt_model_param_info": (c_int, [_P, c_int, POINTER(c_char_p), POINTER(c_int), POINTER(c_int64 * 4), POINTER(c_int)]),
    "umt_model_set_param": (c_int, [_P, c_char_p, _P, c_int, POINTER(c_int64), c_int]),
    "umt_model_finalize": (c_int, [_P, c_int]),
    "umt_model_reserve_memory": (c_int, [_P, c_int, _P]),
    "umt_model_memory_rows": (c_int, [_P, POINTER(c_int)]),
    "umt_model_clear_memory": (c_int, [_P]),
    "umt_model_read_memory": (c_int, [_P, _P, _P, _P]),
    "umt_model_status": (c_int, [_P, POINTER(c_int32), c_int, _P]),
    "umt_model_status_async": (c_int, [_P, _P, c_int, _P]),
    "umt_backbone_workspace_bytes": (c_int64, [c_int]),
    "umt_head_workspace_bytes": (c_int64, [c_int, c_int]),
    "umt_forward_workspace_bytes": (c_int64, [c_int, c_int]),
    "umt_crop_cameras": (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double,
                                 c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "umt_warp_crops": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "umt_resample_pinhole": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "umt_stem_forward": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "umt_backbone_forward": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int64, _P]),
    "umt_fuse_views": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int64, _P]),
    "umt_temporal_step": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int64, _P]),
    "umt_regress_decode": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "umt_model_forward": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int,
                                  _P, _P, _P, _P, _P, c_int64, _P]),
    "umt_head_sequence_workspace_bytes": (c_int64, [c_int, c_int]),
    "umt_head_sequence": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int64, _P]),
    "umt_model_timing_enable": (c_int, [_P, c_int]),
    "umt_model_timing_read": (c_int, [_P, POINTER(c_double), POINTER(c_double), POINTER(c_int)]),
    "umt_model_layer_timing_enable": (c_int, [_P, c_int]),
    "umt_model_layer_timing_read": (c_int, [_P, c_int, _P, POINTER(c_double), POINTER(c_double), POINTER(c_int), POINTER(c_int)]),
    "umt_fk_landmarks": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_float, c_int, _P, _P, _P]),
    "umt_metric_workspace_bytes": (c_int64, [c_int, c_int]),
    "umt_metric_partials": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int64, _P]),
    "umt_metric_partials_static": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
}
METRIC_STATIC_MAX_HAND_FRAMES = 524288

_lib = None


class UmtError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UmtError(f"{LIB_PATH} not found: build it with `make -C umetrack_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)        # AttributeError here = header and library disagree
            fn.restype, fn.argtypes = res, args
        got = handle.umt_abi_version()
        if got != ABI_VERSION:                # a stale .so would be called with mismatched signatures
            raise UmtError(f"{LIB_PATH} reports ABI version {got}, this binding is written for {ABI_VERSION}: "
                           "rebuild it with `make -C umetrack_amd/csrc`")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().umt_last_error().decode("utf-8", "replace")
        if rc == 3:                    # mirrors load_state_dict(strict=True)
            raise KeyError(f"Error(s) in loading state_dict: {msg}")
        raise UmtError(f"{what or 'libumetrack_hip'} failed ({_ERR_NAMES.get(rc, rc)}): {msg}")


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise UmtError("no MI355X visible (torch.cuda.is_available() is False); umetrack_amd has no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise UmtError(f"umetrack_amd runs on gfx950 devices only, got device '{dev}'")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def ptr(t) -> c_void_p:
    """Device (or host) pointer of a contiguous tensor; None -> NULL."""
    if t is None:
        return c_void_p(0)
    assert t.is_contiguous(), "libumetrack_hip needs contiguous buffers"
    return c_void_p(t.data_ptr())


def stream_ptr(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_tensor(t, dtype, device) -> torch.Tensor:
    """Contiguous tensor of `dtype` on `device` (no copy when it already is)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=dtype).contiguous()


class Workspace:
    """Grow-only byte buffer per device, handed to the stage calls (the library never allocates)."""

    def __init__(self):
        self._buf = {}

    def get(self, device, nbytes: int) -> torch.Tensor:
        key = (device.type, device.index)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf
