#!/usr/bin/env python3
"""Round-2 goldens from the imported reference (build container only; the GPU box never runs this).

  python tests/golden/make_goldens_r2.py

1. labels_fixture.json + labels_fixture_ref.npz: a label JSON in the layout video_pose_data.py:72-92 reads
   (cameras given in the three forms read_camera_from_json accepts: {"Camera": {...}}, a bare record, a JSON
   string).  The REFERENCE's lib.common.camera.read_camera_from_json (camera.py:334-355) parses every camera;
   what it built (class, size, f, c, coefficient order) and what those cameras compute (eye_to_window,
   window_to_eye, world_to_eye on fixed points) is stored.  This pins this repo's label reader.
2. eval_tree.npz: three result files written by THIS repo's umetrack_amd.eval._save_result into a protocol
   tree, aggregated by the REFERENCE's load_eval.aggregate_metrics (load_eval.py:47-89, its own pickle.load of
   files this repo wrote); the numbers it prints are parsed and stored next to the arrays.  This pins F4: the
   reference reads our files, and eval.aggregate_metrics must reproduce its MPJPE / AUC / MPJPA.
Only data is written; no reference source is copied.
"""
import contextlib
import io
import json
import os
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def make_labels(rng):
    def fisheye(i):
        rec = {"ImageSizeX": 640, "ImageSizeY": 480, "DistortionModel": "FishEye62",
               "fx": 268.0 + 3 * i, "fy": 269.5 + 2 * i, "cx": 318.2 + i, "cy": 241.7 - i}
        for j, name in enumerate(("k1", "k2", "k3", "k4", "p1", "p2", "k5", "k6")):
            rec[name] = float(rng.uniform(-0.02, 0.02) if name[0] == "k" else rng.uniform(-1e-3, 1e-3)) + 1e-4 * j
        return rec
    pin = {"ImageSizeX": 96, "ImageSizeY": 96, "DistortionModel": "PinholePlane", "fx": 120.0, "fy": 121.0,
           "cx": 47.5, "cy": 47.5}
    cams = [{"Camera": fisheye(0)}, fisheye(1), json.dumps({"Camera": fisheye(2)}), {"Camera": pin}]
    n = 3
    c2w = np.tile(np.eye(4), (n, 4, 1, 1))
    c2w[..., :3, 3] = rng.uniform(-40, 40, size=(n, 4, 3))
    c2w[2] = 0                                                      # an untracked frame (video_pose_data.py:133-137)
    with open(os.path.join(ROOT, "umetrack_amd", "data", "generic_hand_model_small.json")) as fh:
        hm = json.load(fh)
    ja = rng.uniform(-0.3, 0.6, size=(n, 2, 22))
    wr = np.tile(np.eye(4), (n, 2, 1, 1))
    wr[..., :3, 3] = rng.uniform(-50, 50, size=(n, 2, 3)) + [0, 0, 420]
    conf = np.array([[1.0, 1.0], [0.0, 1.0], [0.0, 0.0]])
    return {"cameras": cams, "camera_angles": [0.0, 90.0, -90.0, 180.0], "camera_to_world_transforms": c2w.tolist(),
            "hand_model": hm, "joint_angles": ja.tolist(), "wrist_transforms": wr.tolist(), "hand_confidences": conf.tolist()}


def pin_label_reader():
    from lib.common import camera as ref_camera            # the reference
    rng = np.random.default_rng(2202)
    labels = make_labels(rng)
    with open(os.path.join(HERE, "labels_fixture.json"), "w") as fh:
        json.dump(labels, fh)
    eye_pts = np.concatenate([rng.uniform(-0.6, 0.6, size=(40, 2)), np.ones((40, 1))], -1) * rng.uniform(200, 600, (40, 1))
    out = {"eye_pts": eye_pts}
    for i, c in enumerate(labels["cameras"]):
        cam = ref_camera.read_camera_from_json(c)
        out[f"{i}/cls"] = np.array(type(cam).__name__)
        out[f"{i}/size"] = np.array([cam.width, cam.height])
        out[f"{i}/f"] = np.asarray(cam.f, np.float64)
        out[f"{i}/c"] = np.asarray(cam.c, np.float64)
        out[f"{i}/distort_fields"] = np.array(list(type(cam.distort)._fields))
        out[f"{i}/distort"] = np.asarray(tuple(cam.distort), np.float64)
        out[f"{i}/c2w"] = np.asarray(cam.camera_to_world_xf, np.float64)
        win = cam.eye_to_window(eye_pts)
        out[f"{i}/window"] = win
        if type(cam).__name__ == "PinholePlaneCameraModel":      # window_to_eye asserts a distortion-free camera
            out[f"{i}/rays"] = cam.window_to_eye(win[:, :2])
        moved = cam.copy(camera_to_world_xf=np.asarray(labels["camera_to_world_transforms"])[0, i])
        out[f"{i}/world_to_eye"] = moved.world_to_eye(eye_pts)
    np.savez(os.path.join(HERE, "labels_fixture_ref.npz"), **out)
    print("labels_fixture: %d cameras pinned" % len(labels["cameras"]))


def make_eval_arrays(seed, T):
    rng = np.random.default_rng(seed)
    gt = np.cumsum(rng.normal(size=(2, T, 21, 3)) * 3.0, axis=1) + rng.normal(size=(2, 1, 21, 3)) * 60
    tracked = gt + rng.normal(size=gt.shape) * rng.uniform(2.0, 25.0, size=(2, T, 1, 1))
    valid = rng.uniform(size=(2, T)) > 0.2
    return tracked, gt, valid


EVAL_FILES = (("separate_hand/user01/testing/rec_a.npy", 11, 17), ("separate_hand/user01/testing/rec_b.npy", 12, 5),
              ("separate_hand/user02/testing/rec_c.npy", 13, 40))


def pin_eval_tree():
    import load_eval as ref_load_eval                        # the reference
    from umetrack_amd import eval as our_eval
    out = {"files": np.array([f for f, _, _ in EVAL_FILES])}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (rel, seed, T) in enumerate(EVAL_FILES):
            tracked, gt, valid = make_eval_arrays(seed, T)
            our_eval._save_result(os.path.join(tmp, rel), tracked, gt, valid)
            out[f"{i}/tracked"], out[f"{i}/gt"], out[f"{i}/valid"] = tracked, gt, valid
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_load_eval.aggregate_metrics(tmp)
    text = buf.getvalue()
    print(text)
    num = r"([-+0-9.eE]+)"
    m = re.search(rf"Tracked (\d+) out of (\d+), success rate: {num}%", text)
    out["n_valid"], out["n_total"] = np.int64(m.group(1)), np.int64(m.group(2))
    out["mpjpe"] = np.float64(re.search(rf"Mean keypoint error: {num}", text).group(1))
    out["auc"] = np.float64(re.search(rf"AUC score: {num}", text).group(1))
    out["mpjpa"] = np.float64(re.search(rf"\n  Mean keypoint accelerations: {num}", text).group(1))
    out["gt_mpjpa"] = np.float64(re.search(rf"GT mean keypoint accelerations: {num}", text).group(1))
    np.savez(os.path.join(HERE, "eval_tree.npz"), **out)


if __name__ == "__main__":
    pin_label_reader()
    pin_eval_tree()
