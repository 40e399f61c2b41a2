"""Host side of the network: same call surface as lib/models/umetrack_model.py:21-242.

``UmeTrackModel`` owns an opaque native handle (device weights + temporal state)
and forwards ``regress_pose_use_skeleton`` / ``regress_pose_pred_skel_scale`` to
``umt_model_forward``.  All arithmetic happens in hand-written HIP kernels; the
only torch work here is allocating output and workspace tensors.
"""
import ctypes
import logging
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ... import _native as nat
from ... import arch
from .regressor import RegressorOutput

logger = logging.getLogger(__name__)

_PRECISIONS = {"fp32": nat.PREC_FP32, "bf16": nat.PREC_BF16, "fp32x3": nat.PREC_FP32_BF16X3}
_PRECISIONS_INV = {v: k for k, v in _PRECISIONS.items()}


@dataclass
class InputFrameData:
    """left_images [n,h,w], intrinsics [n,3,3], extrinsics_xf [n,4,4] (umetrack_model.py:21-35)."""
    left_images: torch.Tensor
    intrinsics: torch.Tensor
    extrinsics_xf: torch.Tensor


@dataclass
class InputFrameDesc:
    """sample_range [bs,2], memory_idx [bs], use_memory [bs], hand_idx [bs] (umetrack_model.py:40-61)."""
    sample_range: torch.Tensor
    memory_idx: torch.Tensor
    use_memory: torch.Tensor
    hand_idx: torch.Tensor


@dataclass
class InputSkeletonData:
    """joint_rotation_axes / joint_rest_positions [bs,22,3] or [22,3] (umetrack_model.py:64-74)."""
    joint_rotation_axes: torch.Tensor
    joint_rest_positions: torch.Tensor


class UmeTrackModel:
    """Drop-in for the reference nn.Module (inference only, gfx950 only)."""

    def __init__(self, state_dict: Dict[str, object]):
        L = nat.lib()
        self._h = ctypes.c_void_p()
        nat.check(L.umt_model_create(ctypes.byref(self._h)), "umt_model_create")
        self._device: Optional[torch.device] = None
        self._ws = nat.Workspace()
        self._keep = []
        self._precision = nat.PREC_FP32
        self._load_state_dict(state_dict)

    # ---- load_state_dict(strict=True) semantics, model_loader.py:84-88 ----
    def _load_state_dict(self, sd) -> None:
        L = nat.lib()
        expected = arch.state_spec()
        missing = [k for k in expected if k not in sd]
        unexpected = [k for k in sd if k not in expected]
        if missing or unexpected:
            raise KeyError("Error(s) in loading state_dict: "
                           + (f"Missing key(s): {missing[:5]}{'...' if len(missing) > 5 else ''} " if missing else "")
                           + (f"Unexpected key(s): {unexpected[:5]}" if unexpected else ""))
        for key, (shape, dtype) in expected.items():
            v = sd[key]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(a.shape) != tuple(shape):
                raise KeyError(f"Error(s) in loading state_dict: size mismatch for {key}: "
                               f"{tuple(a.shape)} vs {tuple(shape)}")
            is_int = dtype == "int64"
            a = np.array(a, dtype=np.int64 if is_int else np.float32, order="C")   # keeps 0-d scalars 0-d
            shp = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
            nat.check(L.umt_model_set_param(self._h, key.encode(), a.ctypes.data_as(ctypes.c_void_p), a.ndim,
                                            shp, int(is_int)), f"set_param({key})")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                nat.lib().umt_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- nn.Module-like surface used by the tracker and the eval scripts ----
    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise nat.UmtError("umetrack_amd is inference-only")
        return self

    def to(self, device):
        dev = nat.require_gpu(device)
        if self._device is None:
            with torch.cuda.device(dev):
                nat.check(nat.lib().umt_model_finalize(self._h, dev.index), "umt_model_finalize")
            self._device = dev
        elif dev != self._device:
            raise nat.UmtError(f"model already lives on {self._device}; create one model per GPU")
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    @property
    def device(self) -> Optional[torch.device]:
        return self._device

    # Stated tolerance of the "bf16" mode against the "fp32" mode, per landmark coordinate, in millimetres: 1.5 x the
    # worst of 8 weight seeds on the BASELINE configs[2] batch (profiles/r03_bf16_tolerance_8seeds.json, made by
    # tools/bf16_tolerance.py: per seed mean 0.19-0.68, p99.9 0.81-4.43, max 1.16-8.46 mm).  The cost of the mode depends
    # on the weights; with trained weights it is re-derived by running that script.  tests/test_gpu_bf16.py holds it.
    BF16_LANDMARK_TOLERANCE_MM = {"mean": 1.0, "p999": 6.6, "max": 12.7}

    @property
    def precision(self) -> str:
        """"fp32" (default, the reference's arithmetic), "bf16" (bf16 backbone activations and
        weights with fp32 accumulation; fusion, temporal memory and the head stay fp32) or "fp32x3"
        (fp32 tensors; the stride-1 3x3 backbone layers multiply three-way bf16 splits of their fp32
        operands on the bf16 matrix pipe - fp32-level accuracy, not bit-identical to "fp32").  In both non-fp32 modes the
        known-skeleton regressor's 3x3 layers run on those split-operand kernels too when a forward call has 228 hands or
        more (UMT_HEAD_NOX3 turns that off)."""
        return _PRECISIONS_INV[self._precision]

    @precision.setter
    def precision(self, value: str) -> None:
        if value not in _PRECISIONS:
            raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp32x3', got {value!r}")
        self._precision = _PRECISIONS[value]

    def getInputImageSizes(self) -> Tuple[int, int]:
        return (arch.CROP_HW, arch.CROP_HW)

    def reset_memory(self) -> None:
        """Forget the temporal state (a fresh reference model instance starts like this)."""
        nat.check(nat.lib().umt_model_clear_memory(self._h))

    def memory_state(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Copies of (_mem_features [rows,18,6,6], _prev_extrinsics [rows,4,4]), temporal.py:41-42."""
        L = nat.lib()
        rows = ctypes.c_int()
        nat.check(L.umt_model_memory_rows(self._h, ctypes.byref(rows)))
        mem = torch.empty(rows.value, arch.N_MEM, 6, 6, dtype=torch.float32, device=self._device)
        prev = torch.empty(rows.value, 4, 4, dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            nat.check(L.umt_model_read_memory(self._h, nat.ptr(mem), nat.ptr(prev), nat.stream_ptr(self._device)))
        return mem, prev

    def status(self, clear: bool = True) -> int:
        """The UMT_STATUS_* bits the kernels of this model have raised (one 4-byte copy + stream sync)."""
        word = ctypes.c_int32(0)
        with torch.cuda.device(self._device):
            nat.check(nat.lib().umt_model_status(self._h, ctypes.byref(word), int(clear), nat.stream_ptr(self._device)),
                      "umt_model_status")
        return int(word.value)

    def status_to(self, word: torch.Tensor, clear: bool = True) -> None:
        """Enqueue a copy of the status word into `word` (int32[1] on the model's device) without waiting, so that a
        caller's single device-to-host copy of its outputs carries it along."""
        assert word.is_cuda and word.dtype == torch.int32 and word.numel() == 1
        with torch.cuda.device(self._device):
            nat.check(nat.lib().umt_model_status_async(self._h, nat.ptr(word), int(clear), nat.stream_ptr(self._device)),
                      "umt_model_status_async")

    def check_status(self, raise_nonfinite: bool = False) -> int:
        """Raise where the reference would have raised inside the forward pass (torch.inverse on a singular
        matrix, an index out of range).  A non-finite regressor output (UMT_STATUS_NONFINITE_OUTPUT) is what the
        reference returns as NaN and carries on with: it is logged, and raised only on request.  Returns the bits.
        Call it wherever results reach the host anyway: it synchronises the stream."""
        bits = self.status(clear=True)
        raising = bits if raise_nonfinite else bits & nat.STATUS_RAISING
        if raising:
            raise nat.UmtError("; ".join(msg for bit, msg in nat.STATUS_BITS.items() if raising & bit))
        if bits & nat.STATUS_NONFINITE_OUTPUT:
            logger.warning("a regressor output row was not finite (NaN pose returned, as the reference does)")
        return bits

    def timing_enable(self, on: bool = True) -> None:
        nat.check(nat.lib().umt_model_timing_enable(self._h, int(on)))

    def timing_read(self):
        """(backbone_ms, forward_ms, calls) summed over the forwards since timing_enable(True)."""
        b, f, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        with torch.cuda.device(self._device):
            nat.check(nat.lib().umt_model_timing_read(self._h, ctypes.byref(b), ctypes.byref(f), ctypes.byref(c)))
        return b.value, f.value, c.value

    def layer_timing_enable(self, on: bool = True) -> None:
        nat.check(nat.lib().umt_model_layer_timing_enable(self._h, int(on)))

    def layer_timing_read(self, max_layers: int = 64):
        """Per backbone launch, in schedule order: (name, flop, ms, launches), summed over the passes since
        layer_timing_enable(True); HIP events recorded by the library between the launches."""
        names = ctypes.create_string_buffer(48 * max_layers)
        flop, ms = (ctypes.c_double * max_layers)(), (ctypes.c_double * max_layers)()
        cnt, n = (ctypes.c_int * max_layers)(), ctypes.c_int()
        with torch.cuda.device(self._device):
            nat.check(nat.lib().umt_model_layer_timing_read(self._h, max_layers, names, flop, ms, cnt, ctypes.byref(n)))
        raw = names.raw
        return [(raw[48 * i:48 * (i + 1)].split(b"\0", 1)[0].decode(), flop[i], ms[i], cnt[i]) for i in range(n.value)]

    # ---- forward ----
    def _forward(self, head: int, fd: InputFrameData, desc: InputFrameDesc, skel: Optional[InputSkeletonData],
                 required_rows: Optional[int] = None, out: Optional[RegressorOutput] = None) -> RegressorOutput:
        if self._device is None:
            self.to(fd.left_images.device if fd.left_images.is_cuda else "cuda")
        dev, L = self._device, nat.lib()
        f32, i64 = torch.float32, torch.int64
        images = nat.dev_tensor(fd.left_images, f32, dev)
        if images.dim() != 3 or tuple(images.shape[1:]) != (arch.CROP_HW, arch.CROP_HW):
            raise ValueError(f"left_images must be [n,96,96], got {tuple(images.shape)}")
        n = images.shape[0]
        K = nat.dev_tensor(fd.intrinsics, f32, dev)
        E = nat.dev_tensor(fd.extrinsics_xf, f32, dev)
        sr = nat.dev_tensor(desc.sample_range, i64, dev)
        midx = nat.dev_tensor(desc.memory_idx, i64, dev)
        use = nat.dev_tensor(desc.use_memory, torch.bool, dev).view(torch.uint8)
        hidx = nat.dev_tensor(desc.hand_idx, i64, dev)
        bs = sr.shape[0]
        if tuple(K.shape) != (n, 3, 3) or tuple(E.shape) != (n, 4, 4) or tuple(sr.shape) != (bs, 2) \
                or midx.numel() != bs or use.numel() != bs or hidx.numel() != bs:
            raise ValueError("inconsistent InputFrameData / InputFrameDesc shapes")
        axes = rest = None
        n_skel = 0
        if head == nat.HEAD_KNOWN_SKELETON:
            axes = nat.dev_tensor(skel.joint_rotation_axes, f32, dev).reshape(-1, 22, 3)
            rest = nat.dev_tensor(skel.joint_rest_positions, f32, dev).reshape(-1, 22, 3)
            n_skel = axes.shape[0]
        # temporal.py:102: the memory grows to max(memory_idx)+1 rows (one host sync, as in the reference)
        if required_rows is None:
            required_rows = int(midx.max().item()) + 1
        if out is None:
            out = RegressorOutput(
                joint_angles=torch.empty(bs, 22, dtype=f32, device=dev),
                wrist_xfs=torch.empty(bs, 4, 4, dtype=f32, device=dev),
                skel_scales=torch.empty(bs, dtype=f32, device=dev) if head == nat.HEAD_PREDICT_SCALE else None,
                landmark_uncertainty_sigmas=torch.empty(bs, 21, dtype=f32, device=dev))
        else:       # caller-owned output buffers (e.g. slices of one block that goes to the host in one copy)
            for t, shp in ((out.joint_angles, (bs, 22)), (out.wrist_xfs, (bs, 4, 4)), (out.landmark_uncertainty_sigmas, (bs, 21)),
                           (out.skel_scales, (bs,))):
                if t is None and shp == (bs,) and head != nat.HEAD_PREDICT_SCALE:
                    continue
                if t is None or tuple(t.shape) != shp or t.dtype != f32 or t.device != dev or not t.is_contiguous():
                    raise ValueError(f"out: expected contiguous float32 {shp} on {dev}")
        nbytes = L.umt_forward_workspace_bytes(n, bs)
        ws = self._ws.get(dev, nbytes)
        with torch.cuda.device(dev):
            st = nat.stream_ptr(dev)
            nat.check(L.umt_model_reserve_memory(self._h, required_rows, st), "umt_model_reserve_memory")
            nat.check(L.umt_model_forward(
                self._h, head, self._precision, nat.ptr(images), nat.ptr(K), nat.ptr(E), nat.ptr(sr), nat.ptr(midx),
                nat.ptr(use), nat.ptr(hidx), n, bs, nat.ptr(axes), nat.ptr(rest), n_skel,
                nat.ptr(out.joint_angles), nat.ptr(out.wrist_xfs), nat.ptr(out.landmark_uncertainty_sigmas),
                nat.ptr(out.skel_scales), nat.ptr(ws), ws.numel(), st), "umt_model_forward")
        return out

    def regress_pose_use_skeleton(self, frame_data: InputFrameData, frame_desc: InputFrameDesc,
                                  skel_data: InputSkeletonData, required_rows: Optional[int] = None,
                                  out: Optional[RegressorOutput] = None) -> RegressorOutput:
        """umetrack_model.py:188-219.  `required_rows` (= max(memory_idx) + 1, known to the caller) saves the
        device read-back temporal.py:102 implies; `out` may hold caller-owned output buffers."""
        return self._forward(nat.HEAD_KNOWN_SKELETON, frame_data, frame_desc, skel_data, required_rows, out)

    def regress_pose_pred_skel_scale(self, frame_data: InputFrameData, frame_desc: InputFrameDesc,
                                     required_rows: Optional[int] = None, out: Optional[RegressorOutput] = None,
                                     views_checked: bool = False) -> RegressorOutput:
        """umetrack_model.py:221-242, including its assertion on single-view samples (`views_checked`: the caller
        has made that assertion on its host copy of sample_range, so no device value is read back here)."""
        if not views_checked:
            sr = frame_desc.sample_range
            assert bool(((sr[:, 1] - sr[:, 0]) != 1).all()), \
                "Unsupported: found single-view samples when calibration scale"
        return self._forward(nat.HEAD_PREDICT_SCALE, frame_data, frame_desc, None, required_rows, out)

    # ---- stage-level calls (reference layouts), used by the parity tests ----
    def stem(self, images: torch.Tensor) -> torch.Tensor:
        """The backbone's first layer group alone (conv 1 -> 32, BN, ReLU, 2x2 max-pool; model_utils.py:119-124):
        [n,96,96] -> [n,48,48,32] channels-last, float32, or bfloat16 in the "bf16" mode (the tensor its backbone reads)."""
        dev, L = self._device, nat.lib()
        images = nat.dev_tensor(images, torch.float32, dev)
        n = images.shape[0]
        out = torch.empty(n, 48, 48, 32, device=dev, dtype=torch.bfloat16 if self.precision == "bf16" else torch.float32)
        with torch.cuda.device(dev):
            nat.check(L.umt_stem_forward(self._h, nat.ptr(images), n, self._precision, nat.ptr(out), nat.stream_ptr(dev)),
                      "umt_stem_forward")
        return out

    def backbone(self, images: torch.Tensor) -> torch.Tensor:
        """_image_backbone(images.unsqueeze(1)), umetrack_model.py:127-129: [n,96,96] -> [n,72,6,6]."""
        if self._device is None:
            self.to("cuda")
        dev, L = self._device, nat.lib()
        images = nat.dev_tensor(images, torch.float32, dev)
        if images.dim() != 3 or tuple(images.shape[1:]) != (arch.CROP_HW, arch.CROP_HW):
            raise ValueError(f"left_images must be [n,96,96], got {tuple(images.shape)}")
        n = images.shape[0]
        out = torch.empty(n, 72, 6, 6, dtype=torch.float32, device=dev)
        ws = self._ws.get(dev, L.umt_backbone_workspace_bytes(n))
        with torch.cuda.device(dev):
            nat.check(L.umt_backbone_forward(self._h, nat.ptr(images), n, self._precision, nat.ptr(out), nat.ptr(ws),
                                             ws.numel(), nat.stream_ptr(dev)), "umt_backbone_forward")
        return out

    def fuse_views(self, features, intrinsics, extrinsics, sample_range) -> torch.Tensor:
        """_forward_feature_extractor after the backbone (umetrack_model.py:130-168)."""
        dev, L = self._device, nat.lib()
        f = nat.dev_tensor(features, torch.float32, dev)
        K = nat.dev_tensor(intrinsics, torch.float32, dev)
        E = nat.dev_tensor(extrinsics, torch.float32, dev)
        sr = nat.dev_tensor(sample_range, torch.int64, dev)
        n, bs = f.shape[0], sr.shape[0]
        out = torch.empty(bs, 72, 6, 6, dtype=torch.float32, device=dev)
        ws = self._ws.get(dev, L.umt_head_workspace_bytes(n, bs))
        with torch.cuda.device(dev):
            nat.check(L.umt_fuse_views(self._h, nat.ptr(f), nat.ptr(K), nat.ptr(E), nat.ptr(sr), n, bs, nat.ptr(out),
                                       nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)), "umt_fuse_views")
        return out

    def temporal_step(self, fused, cam0_extrinsics, memory_idx, use_memory,
                      required_rows: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """SimpleConvRNN.forward_temporal_features (temporal.py:93-139)."""
        dev, L = self._device, nat.lib()
        f = nat.dev_tensor(fused, torch.float32, dev)
        E0 = nat.dev_tensor(cam0_extrinsics, torch.float32, dev)
        midx = nat.dev_tensor(memory_idx, torch.int64, dev)
        use = nat.dev_tensor(use_memory, torch.bool, dev).view(torch.uint8)
        bs = f.shape[0]
        if out is None:
            out = torch.empty(bs, 72, 6, 6, dtype=torch.float32, device=dev)
        ws = self._ws.get(dev, L.umt_head_workspace_bytes(1, bs))
        with torch.cuda.device(dev):
            st = nat.stream_ptr(dev)
            rows = int(midx.max().item()) + 1 if required_rows is None else required_rows
            nat.check(L.umt_model_reserve_memory(self._h, rows, st))
            nat.check(L.umt_temporal_step(self._h, nat.ptr(f), nat.ptr(E0), nat.ptr(midx), nat.ptr(use), bs,
                                          nat.ptr(out), nat.ptr(ws), ws.numel(), st), "umt_temporal_step")
        return out

    def head_sequence(self, head: int, fused_t, cam0_t, use_memory_t, hand_idx_t, skel: Optional[InputSkeletonData] = None
                      ) -> RegressorOutput:
        """T lock-step calls of temporal_step + regress_decode in ONE launch (umt_head_sequence): fused_t
        [T,R,72,6,6], cam0_t [T,R,4,4], use_memory_t [T,R] bool, hand_idx_t [T,R]; row r of every step uses memory row
        r.  Returns time-major outputs [T,R,...]; bit-identical to the stepwise calls."""
        dev, L = self._device, nat.lib()
        f32 = torch.float32
        f = nat.dev_tensor(fused_t, f32, dev)
        T, R = int(f.shape[0]), int(f.shape[1])
        E0 = nat.dev_tensor(cam0_t, f32, dev)
        use = nat.dev_tensor(use_memory_t, torch.bool, dev).view(torch.uint8)
        hidx = nat.dev_tensor(hand_idx_t, torch.int64, dev)
        assert tuple(f.shape) == (T, R, 72, 6, 6) and tuple(E0.shape) == (T, R, 4, 4) and use.numel() == T * R \
            and hidx.numel() == T * R
        axes = rest = None
        n_skel = 0
        if head == nat.HEAD_KNOWN_SKELETON:
            axes = nat.dev_tensor(skel.joint_rotation_axes, f32, dev).reshape(-1, 22, 3)
            rest = nat.dev_tensor(skel.joint_rest_positions, f32, dev).reshape(-1, 22, 3)
            n_skel = axes.shape[0]
        out = RegressorOutput(torch.empty(T, R, 22, dtype=f32, device=dev), torch.empty(T, R, 4, 4, dtype=f32, device=dev),
                              torch.empty(T, R, dtype=f32, device=dev) if head == nat.HEAD_PREDICT_SCALE else None,
                              torch.empty(T, R, 21, dtype=f32, device=dev))
        nbytes = L.umt_head_sequence_workspace_bytes(T, R)
        ws = self._ws.get(dev, nbytes)
        with torch.cuda.device(dev):
            st = nat.stream_ptr(dev)
            nat.check(L.umt_model_reserve_memory(self._h, R, st), "umt_model_reserve_memory")
            nat.check(L.umt_head_sequence(self._h, head, nat.ptr(f), nat.ptr(E0), nat.ptr(use), nat.ptr(hidx), nat.ptr(axes),
                                          nat.ptr(rest), n_skel, T, R, nat.ptr(out.joint_angles), nat.ptr(out.wrist_xfs),
                                          nat.ptr(out.landmark_uncertainty_sigmas), nat.ptr(out.skel_scales), nat.ptr(ws),
                                          ws.numel(), st), "umt_head_sequence")
        return out

    def regress_decode(self, head: int, temporal, cam0_extrinsics, hand_idx, skel: Optional[InputSkeletonData] = None,
                       out: Optional[RegressorOutput] = None, want_raw: bool = True):
        """Skeleton encoder + PoseRegressor.regress_poses + wrist recovery; returns (RegressorOutput, raw).
        `out` may hold preallocated contiguous outputs (e.g. slices of a [T, rows, ...] buffer)."""
        dev, L = self._device, nat.lib()
        t = nat.dev_tensor(temporal, torch.float32, dev)
        E0 = nat.dev_tensor(cam0_extrinsics, torch.float32, dev)
        hidx = nat.dev_tensor(hand_idx, torch.int64, dev)
        bs = t.shape[0]
        axes = rest = None
        n_skel = 0
        if head == nat.HEAD_KNOWN_SKELETON:
            axes = nat.dev_tensor(skel.joint_rotation_axes, torch.float32, dev).reshape(-1, 22, 3)
            rest = nat.dev_tensor(skel.joint_rest_positions, torch.float32, dev).reshape(-1, 22, 3)
            n_skel = axes.shape[0]
        f32 = torch.float32
        if out is None:
            out = RegressorOutput(torch.empty(bs, 22, dtype=f32, device=dev), torch.empty(bs, 4, 4, dtype=f32, device=dev),
                                  torch.empty(bs, dtype=f32, device=dev) if head == nat.HEAD_PREDICT_SCALE else None,
                                  torch.empty(bs, 21, dtype=f32, device=dev))
        raw = torch.empty(bs, 62 if head == nat.HEAD_KNOWN_SKELETON else 63, dtype=f32, device=dev) if want_raw else None
        ws = self._ws.get(dev, L.umt_head_workspace_bytes(1, bs))
        with torch.cuda.device(dev):
            nat.check(L.umt_regress_decode(self._h, head, nat.ptr(t), nat.ptr(axes), nat.ptr(rest), n_skel, nat.ptr(E0),
                                           nat.ptr(hidx), bs, nat.ptr(out.joint_angles), nat.ptr(out.wrist_xfs),
                                           nat.ptr(out.landmark_uncertainty_sigmas), nat.ptr(out.skel_scales),
                                           nat.ptr(raw), nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)),
                      "umt_regress_decode")
        return out, raw
