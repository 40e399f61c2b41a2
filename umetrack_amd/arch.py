"""Parameter inventory of the UmeTrack network and seeded synthetic weights.

The reference builds its module tree in lib/models/model_loader.py:53-83 and
loads a plain ``state_dict`` strictly (model_loader.py:84-88).  This module
states the resulting key set as data, so that

* the ctypes host can check a checkpoint strictly before handing it to the
  native library (which holds the same table, see csrc/model.hip), and
* tests / bench can build random-but-deterministic weights without the absent
  ``pretrained_weights.torch``.

Nothing here touches the GPU.
"""
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

# ---- constants of ModelOpts (lib/models/model_opts.py:10-39) ----
STAGE_BLOCKS = (2, 3, 5, 2)           # "resnet_layers_2352"
START_PLANES = 32                      # "-f32"
N_IMG_FEAT = 72
N_SKEL_FEAT = 4
N_MEM = 18
N_WRIST_PTS = 7
FEAT_HW = 6
CROP_HW = 96
BN_EPS = 1e-5

_BB = "_feature_extractor._image_backbone"
_FUSE = "_feature_extractor._multi_view_fusion"
_TEMP = "_temporal._temporal_module"
_SKEL = "_skeleton_enc._layers"

_BN_FIELDS = (("weight", "f"), ("bias", "f"), ("running_mean", "f"),
              ("running_var", "f"), ("num_batches_tracked", "i"))


def _add_bn(spec, prefix, c):
    for name, kind in _BN_FIELDS:
        spec[f"{prefix}.{name}"] = ((), "int64") if kind == "i" else ((c,), "float32")


def _add_conv(spec, prefix, cout, cin, k, bias):
    spec[f"{prefix}.weight"] = ((cout, cin, k, k), "float32")
    if bias:
        spec[f"{prefix}.bias"] = ((cout,), "float32")


def _add_basic_block(spec, prefix, cin, cout, downsample):
    # backbone_resnet.py:22-54 ; downsample at backbone_resnet.py:134-146
    _add_conv(spec, f"{prefix}.conv1", cout, cin, 3, False)
    _add_bn(spec, f"{prefix}.bn1", cout)
    _add_conv(spec, f"{prefix}.conv2", cout, cout, 3, False)
    _add_bn(spec, f"{prefix}.bn2", cout)
    if downsample:
        _add_conv(spec, f"{prefix}.downsample.0", cout, cin, 1, False)
        _add_bn(spec, f"{prefix}.downsample.1", cout)


def stage_planes():
    """(in_planes, out_planes, stride) per ResNet stage, backbone_resnet.py:161-166."""
    p = START_PLANES
    return ((p, p, 1), (p, 2 * p, 2), (2 * p, 4 * p, 2), (4 * p, 8 * p, 2))


def regressor_out_dims(predict_scale: bool) -> int:
    # regressor.py:61-73: 20 angles + 7*3 wrist points [+1 scale] + 21 sigmas
    return 20 + 3 * N_WRIST_PTS + (1 if predict_scale else 0) + 21


def regressor_slices(predict_scale: bool) -> Dict[str, Tuple[int, int]]:
    out, n = OrderedDict(), 0
    for key, width in (("joint_angles", 20), ("wrist_xfs", 3 * N_WRIST_PTS),
                       ("skel_scales", 1 if predict_scale else 0),
                       ("landmark_uncertainty_sigmas", 21)):
        if width:
            out[key] = (n, n + width)
            n += width
    return out


def state_spec() -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """key -> (shape, dtype) in the reference's state_dict order."""
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    # stem: model_utils.py:119-124
    _add_conv(s, f"{_BB}.0._layers.0.0", START_PLANES, 1, 3, True)
    _add_bn(s, f"{_BB}.0._layers.0.1", START_PLANES)
    for si, ((cin, cout, stride), nblk) in enumerate(zip(stage_planes(), STAGE_BLOCKS), 1):
        for b in range(nblk):
            first = b == 0
            _add_basic_block(s, f"{_BB}.0._layers.{si}.{b}", cin if first else cout, cout,
                             first and (stride != 1 or cin != cout))
    _add_conv(s, f"{_BB}.1", N_IMG_FEAT, 8 * START_PLANES, 1, True)      # model_utils.py:134
    # fusion: model_utils.py:141-163, linspace(144, 72, 3) -> 144,108,72
    _add_conv(s, f"{_FUSE}.0", 108, 144, 1, True)
    _add_bn(s, f"{_FUSE}.1", 108)
    _add_conv(s, f"{_FUSE}.3", 72, 108, 1, True)
    _add_bn(s, f"{_FUSE}.4", 72)
    _add_conv(s, f"{_FUSE}.6", 72, 72, 1, True)
    # temporal: temporal.py:29-38
    nt = N_IMG_FEAT + N_MEM
    for i in (0, 2, 4):
        _add_conv(s, f"{_TEMP}.{i}", nt, nt, 1, True)
    # skeleton encoder: skeleton_encoder.py:36-41
    s[f"{_SKEL}.0.weight"] = ((N_SKEL_FEAT * FEAT_HW * FEAT_HW, 22 * 6), "float32")
    s[f"{_SKEL}.0.bias"] = ((N_SKEL_FEAT * FEAT_HW * FEAT_HW,), "float32")
    _add_bn(s, f"{_SKEL}.2", N_SKEL_FEAT)
    # regressors: model_utils.py:195-208, model_loader.py:64-75
    for name, cin, pscale in (("_regressor_k", N_IMG_FEAT + N_SKEL_FEAT, False),
                              ("_regressor_u", N_IMG_FEAT, True)):
        p = f"{name}._pose_regression_layers"
        for b in range(2):
            _add_basic_block(s, f"{p}.{b}", cin, cin, False)
        _add_conv(s, f"{p}.2", regressor_out_dims(pscale), cin, 1, True)
    return s


def make_synthetic_state(seed: int = 0, out_gain: float = 0.02) -> "OrderedDict[str, np.ndarray]":
    """Seeded random weights with non-trivial BN statistics.

    numpy's PCG64 stream is stable across versions, so the same seed gives the
    same weights here, in the golden generator and on the GPU box.  Convs use
    the reference's He initialisation std (backbone_resnet.py:117-120); BN
    gamma/beta/mean/var are randomised because the stock init leaves BN at
    identity which would not exercise the BN epilogue.
    ``out_gain`` scales the last conv of each regressor so decoded wrist points
    land at tens of centimetres, like real poses.
    """
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, (shape, dtype) in state_spec().items():
        leaf = key.rsplit(".", 1)[1]
        if dtype == "int64":
            out[key] = np.array(1, dtype=np.int64)
        elif len(shape) == 4:
            cout, _, k, _ = shape
            std = np.sqrt(2.0 / (k * k * cout))
            if key.endswith("_pose_regression_layers.2.weight"):
                std *= out_gain
            out[key] = (rng.standard_normal(shape) * std).astype(np.float32)
        elif len(shape) == 2:
            out[key] = (rng.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
        elif leaf == "running_var":
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "running_mean":
            out[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "weight":          # BN gamma
            out[key] = rng.uniform(0.6, 1.1, shape).astype(np.float32)
        else:                            # conv / linear / BN bias
            out[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
    return out
