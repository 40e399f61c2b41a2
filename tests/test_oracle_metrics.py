"""Pin oracle/metrics.py against load_eval.py / metric_utils.py outputs."""
import numpy as np

from conftest import load_npz
from oracle import metrics as omx


def test_metrics_match_reference():
    g = load_npz("metrics.npz")
    recs = []
    for r in range(int(g["n_rec"])):
        gt, pred, valid = g[f"{r}/gt"], g[f"{r}/pred"], g[f"{r}/valid"]
        e, a, ga = omx.per_recording(gt, pred, valid)
        np.testing.assert_allclose(e, g[f"{r}/err"], rtol=1e-13)
        np.testing.assert_allclose(a, g[f"{r}/acc"], rtol=1e-13)
        np.testing.assert_allclose(ga, g[f"{r}/gt_acc"], rtol=1e-13)
        recs.append((gt, pred, valid))
    agg = omx.aggregate(recs)
    np.testing.assert_allclose(agg["mpjpe"], float(g["mpjpe"]), rtol=1e-13)
    np.testing.assert_allclose(agg["mpjpa"], float(g["mpjpa"]), rtol=1e-13)
    np.testing.assert_allclose(agg["gt_mpjpa"], float(g["gt_mpjpa"]), rtol=1e-13)
    np.testing.assert_allclose(agg["auc"], float(g["auc"]), rtol=1e-12)
    np.testing.assert_allclose(agg["pck_counts"] / agg["n_err"] * 100.0, g["pck"], rtol=1e-12)
