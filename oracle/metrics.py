"""CPU restatement of the evaluation metrics (row A15).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows load_eval.py:26-89 and
lib/common/metric_utils.py:18-112.  Pinned against tests/golden/metrics.npz
(the reference's own functions on random arrays).
"""
import numpy as np

PCK_THRESHOLDS = np.linspace(0, 50, 101)          # metric_utils.py:18-19


def per_recording(gt: np.ndarray, pred: np.ndarray, valid: np.ndarray):
    """load_eval.py:26-44.  gt/pred [2,T,21,3], valid [2,T] bool."""
    err = np.linalg.norm(gt - pred, axis=-1).mean(-1)

    def acc(p):
        return np.linalg.norm(p[:, :-2] + p[:, 2:] - 2 * p[:, 1:-1], axis=-1).mean(-1)

    v3 = valid[:, :-2] & valid[:, 1:-1] & valid[:, 2:]
    return err[valid], acc(pred)[v3], acc(gt)[v3]


def pck_curve(errors: np.ndarray, thresholds: np.ndarray = PCK_THRESHOLDS) -> np.ndarray:
    """metric_utils.py:35-84 with mask = ones, axis = None."""
    e = errors.reshape(-1)
    n = e.size
    return np.array([((e <= t).sum() / n) if n >= 1e-6 else 0 for t in thresholds])


def normalized_auc(x: np.ndarray, y: np.ndarray, y_max: float = 1.0) -> float:
    """metric_utils.py:87-112 (trapezoid rule / full area)."""
    return float(((x[1:] - x[:-1]) * (y[1:] + y[:-1]) * 0.5).sum() / ((x[-1] - x[0]) * y_max))


def aggregate(recordings):
    """load_eval.py:47-89 over a list of (gt, pred, valid) triples."""
    errs, accs, gaccs, n_valid, n_total = [], [], [], 0, 0
    for gt, pred, valid in recordings:
        e, a, g = per_recording(gt, pred, valid)
        errs.append(e)
        accs.append(a)
        gaccs.append(g)
        n_valid += int(valid.sum())
        n_total += valid.size
    e, a, g = np.concatenate(errs), np.concatenate(accs), np.concatenate(gaccs)
    pck = pck_curve(e) * 100.0
    return {
        "mpjpe": float(e.mean()), "mpjpa": float(a.mean()), "gt_mpjpa": float(g.mean()),
        "auc": normalized_auc(PCK_THRESHOLDS, pck), "n_valid": n_valid, "n_total": n_total,
        "sum_err": float(e.sum()), "n_err": int(e.size), "sum_acc": float(a.sum()),
        "n_acc": int(a.size), "sum_gt_acc": float(g.sum()),
        "pck_counts": np.array([(e <= t).sum() for t in PCK_THRESHOLDS], dtype=np.int64),
    }
