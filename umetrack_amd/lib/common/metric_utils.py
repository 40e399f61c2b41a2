"""Evaluation metrics: device partial sums + the host-side curve maths.

Same constants and function names as lib/common/metric_utils.py:18-112; the
per-frame work of load_eval._compute_metrics (load_eval.py:26-44) runs in
csrc/metrics.hip (umt_metric_partials).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from ... import _native as nat

MAX_LANDMARK_ERROR_MM = 50
PCK_THRESHOLDS = np.linspace(0, MAX_LANDMARK_ERROR_MM, 101)
N_SUMS = 7          # sum_err, n_err, sum_acc, n_acc, sum_gt_acc, n_valid, n_total


def metric_partials(gt: torch.Tensor, pred: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """gt/pred [n_rec,2,T,21,3] (or [2,T,21,3]), valid [n_rec,2,T] -> float64[108] on the GPU."""
    dev = nat.require_gpu(gt.device if gt.is_cuda else None)
    if gt.dim() == 4:
        gt, pred, valid = gt[None], pred[None], valid[None]
    g = nat.dev_tensor(gt, torch.float32, dev)
    p = nat.dev_tensor(pred, torch.float32, dev)
    v = nat.dev_tensor(valid, torch.bool, dev).view(torch.uint8)
    n_rec, two, T = v.shape
    assert two == 2 and tuple(g.shape) == (n_rec, 2, T, 21, 3) and g.shape == p.shape
    L = nat.lib()
    out = torch.empty(nat.METRIC_WORDS, dtype=torch.float64, device=dev)
    nbytes = L.umt_metric_workspace_bytes(n_rec, T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.umt_metric_partials(nat.ptr(g), nat.ptr(p), nat.ptr(v), n_rec, T, nat.ptr(out), nat.ptr(ws),
                                        nbytes, nat.stream_ptr(dev)), "umt_metric_partials")
    return out


@dataclass
class AggregateMetrics:
    mpjpe: float
    mpjpa: float
    gt_mpjpa: float
    auc: float
    n_valid: int
    n_total: int
    pck: np.ndarray


def finalize_partials(partials) -> AggregateMetrics:
    """Turn (summed) partials into the numbers aggregate_metrics prints (load_eval.py:66-89)."""
    p = partials.detach().cpu().numpy() if isinstance(partials, torch.Tensor) else np.asarray(partials)
    sum_err, n_err, sum_acc, n_acc, sum_gacc, n_valid, n_total = p[:N_SUMS]
    pck = (p[N_SUMS:] / n_err if n_err >= 1e-6 else np.zeros(101)) * 100.0
    return AggregateMetrics(mpjpe=float(sum_err / n_err) if n_err else float("nan"),
                            mpjpa=float(sum_acc / n_acc) if n_acc else float("nan"),
                            gt_mpjpa=float(sum_gacc / n_acc) if n_acc else float("nan"),
                            auc=float(normalized_AUC(PCK_THRESHOLDS, pck)),
                            n_valid=int(n_valid), n_total=int(n_total), pck=pck)


def PCK_curve(errors: np.ndarray, thresholds: np.ndarray, mask: Optional[np.ndarray] = None,
              axis: Optional[int] = None) -> np.ndarray:
    """Host-side PCK for already computed errors (metric_utils.py:46-84)."""
    if mask is None:
        mask = np.ones_like(errors)
    if axis is None:
        e, m = errors.reshape(1, -1), mask.reshape(1, -1)
    else:
        n = errors.shape[axis]
        e, m = np.moveaxis(errors, axis, 0).reshape(n, -1), np.moveaxis(mask, axis, 0).reshape(n, -1)
    den = m.sum(-1)
    cols = []
    for t in thresholds:
        num = ((e <= t) * m).sum(-1)
        cols.append(np.where(den < 1e-6, 0.0, num / np.where(den < 1e-6, 1.0, den)))
    out = np.stack(cols).T
    return out[0] if axis is None else out


def normalized_AUC(x: np.ndarray, y: np.ndarray, y_max: float = 1.0) -> np.ndarray:
    """Trapezoid area under y(x) over the full rectangle (metric_utils.py:87-112)."""
    lead = y.shape[:-1]
    y2 = y.reshape(-1, y.shape[-1])
    area = (np.diff(x)[None, :] * 0.5 * (y2[:, 1:] + y2[:, :-1])).sum(-1)
    return (area / ((x[-1] - x[0]) * y_max)).reshape(lead)
