"""Fixtures made by tests/golden/make_goldens_r2.py from the imported reference:
the label-JSON reader against the reference's read_camera_from_json (camera.py:334-355), the result tree read by
the reference's load_eval.aggregate_metrics (load_eval.py:47-89), and the restricted result-file loader."""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN, load_npz
from oracle import metrics as omx


def test_label_reader_matches_reference_cameras():
    from umetrack_amd.lib.common import camera
    from umetrack_amd.lib.tracker.video_pose_data import _load_hand_pose_labels
    ref = load_npz("labels_fixture_ref.npz")
    labels = _load_hand_pose_labels(os.path.join(GOLDEN, "labels_fixture.json"))
    assert len(labels.cameras) == 4 and len(labels) == 3
    assert labels.camera_angles == [0.0, 90.0, -90.0, 180.0]
    eye = ref["eye_pts"]
    for i, cam in enumerate(labels.cameras):
        assert type(cam).__name__ == str(ref[f"{i}/cls"])
        assert [cam.width, cam.height] == ref[f"{i}/size"].tolist()
        np.testing.assert_array_equal(np.asarray(cam.f, np.float64), ref[f"{i}/f"])
        np.testing.assert_array_equal(np.asarray(cam.c, np.float64), ref[f"{i}/c"])
        assert list(type(cam.distort)._fields) == ref[f"{i}/distort_fields"].tolist()          # k1..k4, p1, p2, k5, k6
        np.testing.assert_array_equal(np.asarray(tuple(cam.distort), np.float64), ref[f"{i}/distort"])
        np.testing.assert_array_equal(np.asarray(cam.camera_to_world_xf, np.float64), ref[f"{i}/c2w"])
        np.testing.assert_allclose(cam.eye_to_window(eye), ref[f"{i}/window"], rtol=1e-12, atol=1e-9)
        if f"{i}/rays" in ref:
            np.testing.assert_allclose(cam.window_to_eye(ref[f"{i}/window"][:, :2]), ref[f"{i}/rays"], rtol=1e-12, atol=1e-12)
        moved = cam.copy(camera_to_world_xf=labels.camera_to_world_transforms[0, i])
        np.testing.assert_allclose(moved.world_to_eye(eye), ref[f"{i}/world_to_eye"], rtol=1e-12, atol=1e-9)
    assert isinstance(labels.cameras[3], camera.PinholePlaneCameraModel)
    assert labels.joint_angles.shape == (3, 2, 22) and labels.wrist_transforms.shape == (3, 2, 4, 4)
    assert labels.hand_confidences.tolist() == [[1.0, 1.0], [0.0, 1.0], [0.0, 0.0]]
    assert labels.camera_to_world_transforms[2].sum() == 0
    assert tuple(labels.hand_model.joint_rest_positions.shape) == (22, 3)


def _eval_tree():
    g = load_npz("eval_tree.npz")
    recs = [(g[f"{i}/gt"], g[f"{i}/tracked"], g[f"{i}/valid"]) for i in range(len(g["files"]))]
    return g, recs


def test_oracle_aggregate_matches_reference_on_result_tree():
    g, recs = _eval_tree()
    agg = omx.aggregate(recs)
    assert agg["n_valid"] == int(g["n_valid"]) and agg["n_total"] == int(g["n_total"])
    for key in ("mpjpe", "auc", "mpjpa", "gt_mpjpa"):
        np.testing.assert_allclose(agg[key], float(g[key]), rtol=1e-13, err_msg=key)


def test_result_files_round_trip_and_reject_foreign_pickles(tmp_path):
    from umetrack_amd import eval as ev
    g, recs = _eval_tree()
    path = tmp_path / "a" / "rec.npy"
    gt, tracked, valid = recs[0]
    ev._save_result(str(path), tracked, gt, valid)
    with open(path, "rb") as fp:                      # what the reference does with it (load_eval.py:55-56)
        plain = pickle.load(fp)
    d = ev.load_result(str(path))
    for key in ("tracked_keypoints", "gt_keypoints", "valid_tracking"):
        np.testing.assert_array_equal(d[key], plain[key])
    assert d["valid_tracking"].dtype == bool and d["gt_keypoints"].dtype == np.float64

    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, (f"touch {marker}",))

    bad = tmp_path / "evil.npy"
    with open(bad, "wb") as fp:
        pickle.dump({"tracked_keypoints": Evil(), "gt_keypoints": gt, "valid_tracking": valid}, fp)
    with pytest.raises(pickle.UnpicklingError, match="only numpy arrays"):
        ev.load_result(str(bad))
    assert not marker.exists()
    # a well-formed pickle of the wrong shape of thing is refused too
    odd = tmp_path / "odd.npy"
    with open(odd, "wb") as fp:
        pickle.dump({"tracked_keypoints": tracked, "gt_keypoints": gt[:, :3], "valid_tracking": valid}, fp)
    with pytest.raises(pickle.UnpicklingError, match="inconsistent"):
        ev.load_result(str(odd))
    with open(odd, "wb") as fp:
        pickle.dump([1, 2, 3], fp)
    with pytest.raises(pickle.UnpicklingError, match="expected a dict"):
        ev.load_result(str(odd))


@pytest.mark.gpu
def test_gpu_aggregate_reproduces_reference_numbers_on_result_tree(tmp_path):
    """F4 both ways: the reference's load_eval read a tree written by _save_result (that is how eval_tree.npz was
    made); here the same tree is written again and eval.aggregate_metrics (GPU partial sums) must print the
    reference's MPJPE / AUC / MPJPA to 2 decimals."""
    from umetrack_amd import eval as ev
    g, recs = _eval_tree()
    for rel, (gt, tracked, valid) in zip(g["files"].tolist(), recs):
        ev._save_result(str(tmp_path / rel), tracked, gt, valid)
    m = ev.aggregate_metrics(str(tmp_path), verbose=False)
    assert m.n_valid == int(g["n_valid"]) and m.n_total == int(g["n_total"])
    for key in ("mpjpe", "auc", "mpjpa", "gt_mpjpa"):
        got, want = getattr(m, key), float(g[key])
        assert abs(got - want) < 5e-4 * max(1.0, abs(want)), (key, got, want)      # fp32 landmarks on the device
        assert f"{got:.2f}" == f"{want:.2f}", (key, got, want)
