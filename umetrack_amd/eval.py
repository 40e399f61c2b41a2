"""Evaluation loops with the reference's on-disk formats.

* ``track_sequence`` / ``track_sequence_unknown_skeleton`` follow
  run_eval_known_skeleton.py:48-102 and run_eval_unknown_skeleton.py:49-144 over any stream that
  yields ``(InputFrame, gt_tracking)`` like lib/tracker/video_pose_data.py:106-151 and exposes
  ``hand_model`` and ``camera_angles``.  Landmarks are produced by the GPU FK kernel.
* results are written exactly as the reference writes them: a pickled dict with
  ``tracked_keypoints`` / ``gt_keypoints`` ``[2,T,21,3]`` float64 and ``valid_tracking`` ``[2,T]``
  bool under a ``.npy`` name (run_eval_known_skeleton.py:90-100), so the reference's own
  load_eval.py can read them.
* ``aggregate_metrics`` is load_eval.py:47-89 with the per-frame arithmetic on the GPU and an
  optional all-reduce over ranks (recordings are sharded whole, see umetrack_amd.dist).

The mp4 / label-JSON reader itself (needs ``av``) is outside this package; ``SyntheticRecording``
stands in for it in tests and benchmarks.
"""
import fnmatch
import logging
import os
import pickle
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist as udist
from . import synthetic
from .lib.common import camera
from .lib.common import metric_utils as mu
from .lib.common.hand import NUM_HANDS, NUM_LANDMARKS_PER_HAND, HandModel, scaled_hand_model
from .lib.tracker.perspective_crop import landmarks_from_hand_pose
from .lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
from .lib.tracker.tracking_result import SingleHandPose

logger = logging.getLogger(__name__)


class SyntheticRecording:
    """In-memory stand-in for SyncedImagePoseStream: random fisheye frames, random poses.

    Frames are one uint8 array [H, 4*W] with the views side by side, split exactly as
    video_pose_data.py:123-125 does."""

    def __init__(self, n_frames: int, seed: int, hand_model: HandModel, drop_left_every: int = 0):
        rng = np.random.default_rng(seed)
        intr, coeffs, c2w = synthetic.make_rig(rng)
        self.cameras = [camera.Fisheye62CameraModel(synthetic.SRC_W, synthetic.SRC_H, tuple(intr[v, :2]),
                                                    tuple(intr[v, 2:]), list(coeffs[v]), c2w[v]) for v in range(4)]
        self.camera_angles = [0.0] * 4
        self.hand_model = hand_model
        lim = np.asarray(hand_model.joint_limits, np.float64)
        self._frames = rng.integers(0, 256, size=(n_frames, synthetic.SRC_H, synthetic.SRC_W * 4), dtype=np.uint8)
        ja = rng.uniform(lim[:, 0], lim[:, 1], size=(n_frames, 2, 22)).astype(np.float32)
        ja[..., 20:] = 0
        wr = np.tile(np.eye(4, dtype=np.float32), (n_frames, 2, 1, 1))
        wr[..., :3, :3] = synthetic.random_rotations(rng, n_frames * 2).reshape(n_frames, 2, 3, 3)
        wr[..., :3, 3] = np.stack([rng.uniform(-60, 60, (n_frames, 2)), rng.uniform(-40, 40, (n_frames, 2)),
                                   rng.uniform(380, 480, (n_frames, 2))], -1)
        self.joint_angles, self.wrist_transforms = ja, wr
        conf = np.ones((n_frames, 2), np.float32)
        if drop_left_every:
            conf[drop_left_every - 1::drop_left_every, 0] = 0
        self.hand_confidences = conf

    def __len__(self):
        return self._frames.shape[0]

    def __iter__(self):
        for f in range(len(self)):
            gt = {h: SingleHandPose(self.joint_angles[f, h], self.wrist_transforms[f, h], float(self.hand_confidences[f, h]))
                  for h in range(2) if self.hand_confidences[f, h] > 0}
            mv = self._frames[f].reshape(synthetic.SRC_H, 4, -1)
            views = [ViewData(image=mv[:, v, :], camera=self.cameras[v], camera_angle=self.camera_angles[v])
                     for v in range(4)]
            yield InputFrame(views=views), gt


def _save_result(output_path: Optional[str], tracked, gt, valid) -> None:
    if output_path is None:
        return
    os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
    with open(output_path, "wb") as fp:
        pickle.dump({"tracked_keypoints": tracked, "gt_keypoints": gt, "valid_tracking": valid}, fp)
    logger.info("Results saved at %s", output_path)


def _track(stream, tracker: HandTracker, track_model: HandModel, gt_model: HandModel):
    n = len(stream)
    gt_kp = np.zeros([NUM_HANDS, n, NUM_LANDMARKS_PER_HAND, 3])
    tr_kp = np.zeros([NUM_HANDS, n, NUM_LANDMARKS_PER_HAND, 3])
    valid = np.zeros([NUM_HANDS, n], dtype=bool)
    for frame_idx, (frame, gt_tracking) in enumerate(stream):
        crop_cameras = tracker.gen_crop_cameras([v.camera for v in frame.views], stream.camera_angles, gt_model,
                                                gt_tracking, min_num_crops=1)
        res = tracker.track_frame(frame, track_model, crop_cameras)
        for hand_idx in res.hand_poses:
            tr_kp[hand_idx, frame_idx] = landmarks_from_hand_pose(track_model, res.hand_poses[hand_idx], hand_idx)
            gt_kp[hand_idx, frame_idx] = landmarks_from_hand_pose(gt_model, gt_tracking[hand_idx], hand_idx)
            valid[hand_idx, frame_idx] = True
    return tr_kp, gt_kp, valid


def track_sequence(stream, model, output_path: Optional[str] = None, override: bool = False):
    """run_eval_known_skeleton.py:48-102.  Returns the per-frame mean errors (mm) of valid hands,
    or None when the output already exists (the reference's resume rule)."""
    if output_path is not None and not override and os.path.exists(output_path):
        logger.info("Skipping since output path '%s' already exists", output_path)
        return None
    model.reset_memory()                                   # the reference builds a fresh model per recording
    tracker = HandTracker(model, HandTrackerOpts())
    tr_kp, gt_kp, valid = _track(stream, tracker, stream.hand_model, stream.hand_model)
    _save_result(output_path, tr_kp, gt_kp, valid)
    return np.linalg.norm((gt_kp - tr_kp)[valid], axis=-1).mean(axis=-1)


def calibrate_scale(stream, tracker: HandTracker, generic_hand_model: HandModel, n_calibration_samples: int = 30):
    """run_eval_unknown_skeleton.py:49-78: mean of the first predicted scales from 2-view frames."""
    samples: List[float] = []
    for frame, gt_tracking in stream:
        crop_cameras = tracker.gen_crop_cameras([v.camera for v in frame.views], stream.camera_angles,
                                                stream.hand_model, gt_tracking, min_num_crops=2)
        res = tracker.track_frame_and_calibrate_scale(frame, crop_cameras)
        samples.extend(float(res.predicted_scales[h]) for h in res.hand_poses)
        if n_calibration_samples and len(samples) >= n_calibration_samples:
            samples = samples[:n_calibration_samples]
            break
    assert samples, "No samples collected for scale calibration!"
    mean_scale = float(np.mean(samples))
    logger.info("Calibrated mean scale: %s with %d samples", mean_scale, len(samples))
    return scaled_hand_model(generic_hand_model, mean_scale), mean_scale


def track_sequence_unknown_skeleton(stream, model, generic_hand_model: HandModel, n_calibration_samples: int = 30,
                                    output_path: Optional[str] = None, override: bool = False):
    """run_eval_unknown_skeleton.py:81-144."""
    if output_path is not None and not override and os.path.exists(output_path):
        return None
    model.reset_memory()
    tracker = HandTracker(model, HandTrackerOpts())
    calibrated, _ = calibrate_scale(stream, tracker, generic_hand_model, n_calibration_samples)
    tracker.reset_history()
    tr_kp, gt_kp, valid = _track(stream, tracker, calibrated, stream.hand_model)
    _save_result(output_path, tr_kp, gt_kp, valid)
    return np.linalg.norm((gt_kp - tr_kp)[valid], axis=-1).mean(axis=-1)


def find_result_files(output_dir: str) -> List[str]:
    out = []
    for cur, _, names in os.walk(output_dir):
        out += [os.path.join(cur, n) for n in fnmatch.filter(names, "*.npy")]
    return sorted(out)


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Result files are pickles of a dict of three numpy arrays (run_eval_known_skeleton.py:92-100).  A pickle
    can name any callable; this loader resolves only what numpy arrays need and refuses everything else, so a
    tampered or foreign result file cannot execute code."""
    _ALLOWED = {("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError(f"result file refers to {module}.{name}: only numpy arrays are accepted")
        return super().find_class(module, name)


def load_result(path: str) -> Dict[str, np.ndarray]:
    """Result files are this package's own output or the reference scripts'; like load_eval.py:55-56 they are
    pickles despite the .npy suffix.  Loaded with an unpickler restricted to numpy arrays; the three fields are
    checked for type and shape."""
    with open(path, "rb") as fp:
        d = _ArraysOnlyUnpickler(fp).load()
    if not isinstance(d, dict):
        raise pickle.UnpicklingError(f"{path}: expected a dict of arrays, got {type(d).__name__}")
    for key in ("tracked_keypoints", "gt_keypoints", "valid_tracking"):
        if not isinstance(d.get(key), np.ndarray):
            raise pickle.UnpicklingError(f"{path}: field {key!r} is missing or not a numpy array")
    if d["tracked_keypoints"].shape != d["gt_keypoints"].shape or d["valid_tracking"].shape != d["gt_keypoints"].shape[:2]:
        raise pickle.UnpicklingError(f"{path}: inconsistent array shapes")
    return d


def aggregate_metrics(output_dir: str, files: Optional[Sequence[str]] = None, reduce_over_ranks: bool = False,
                      verbose: bool = True) -> Optional[mu.AggregateMetrics]:
    """load_eval.py:47-89 with the sums computed by the GPU metric kernel.  With
    ``reduce_over_ranks`` every rank passes its own ``files`` and all ranks get the global result."""
    if files is None:
        files = find_result_files(output_dir)
    total = None
    for f in files:
        d = load_result(f)
        if d["valid_tracking"].shape[1] == 0:
            continue
        part = mu.metric_partials(torch.from_numpy(d["gt_keypoints"]), torch.from_numpy(d["tracked_keypoints"]),
                                  torch.from_numpy(d["valid_tracking"]))
        total = part.clone() if total is None else total + part.to(total.device)    # lives where the partials live
    if total is None:
        # a rank without files still takes part in the all-reduce, with zeros on the device the collective runs on:
        # the GPU for RCCL ("nccl"), the host for gloo
        on_gpu = (reduce_over_ranks and torch.distributed.is_available() and torch.distributed.is_initialized()
                  and torch.distributed.get_backend() == "nccl")
        total = torch.zeros(108, dtype=torch.float64, device="cuda" if on_gpu else "cpu")
    if reduce_over_ranks:
        udist.all_reduce_partials(total)
    if float(total[1]) == 0:
        return None
    m = mu.finalize_partials(total)
    if verbose:
        print(f"  Tracked {m.n_valid} out of {m.n_total}, success rate: {m.n_valid / m.n_total * 100}%")
        print(f"  Mean keypoint error: {m.mpjpe}")
        print(f"  AUC score: {m.auc}")
        print(f"  Mean keypoint accelerations: {m.mpjpa}")
        print(f"  GT mean keypoint accelerations: {m.gt_mpjpa}")
    return m


# ---- whole-dataset drivers: run_eval_known_skeleton.py:100-119, run_eval_unknown_skeleton.py:145-172 ----
def find_input_output_files(input_dir: str, output_dir: str, test_only: bool,
                            patterns: Sequence[str] = ("*.mp4", "*.mono")):
    """run_eval_known_skeleton.py:31-45: every recording under ``input_dir`` (only folders whose path
    contains "testing" when ``test_only``) and its result path ``output_dir/<relative path>.npy``.
    .mono recordings (lib/tracker/video_pose_data.py) are found next to .mp4 ones."""
    ins, outs = [], []
    for cur, _, names in sorted(os.walk(input_dir)):
        if test_only and "testing" not in cur:
            continue
        for pat in patterns:
            for name in sorted(fnmatch.filter(names, pat)):
                full = os.path.join(cur, name)
                rel = os.path.relpath(full, input_dir)
                ins.append(full)
                outs.append(os.path.join(output_dir, os.path.splitext(rel)[0] + ".npy"))
    logger.info("Found %d files from %s", len(ins), input_dir)
    return ins, outs


def run_eval(input_dir: str, output_dir: str, model, protocol: str = "known", generic_hand_model: Optional[HandModel] = None,
             n_calibration_samples: int = 30, test_only: bool = True, rank: int = 0, world: int = 1,
             override: bool = False):
    """Track every recording of a dataset.  The reference fans recordings out over ``Pool(8)`` CPU
    processes; here each rank (one per GPU, torchrun) takes whole recordings, balanced by size, and
    ``aggregate_metrics(..., reduce_over_ranks=True)`` combines the result with one all-reduce.
    Returns (this rank's per-frame errors concatenated, its result files)."""
    from .lib.tracker.video_pose_data import SyncedImagePoseStream
    if protocol not in ("known", "unknown"):
        raise ValueError(f"protocol must be 'known' or 'unknown', got {protocol!r}")
    if protocol == "unknown" and generic_hand_model is None:
        raise ValueError("the unknown-skeleton protocol needs the generic hand model")
    ins, outs = find_input_output_files(input_dir, output_dir, test_only)
    mine = udist.shard_recordings([os.path.getsize(p) for p in ins], world)[rank]
    errors, files = [], []
    for i in mine:
        stream = SyncedImagePoseStream(ins[i])
        if protocol == "known":
            e = track_sequence(stream, model, outs[i], override)
        else:
            e = track_sequence_unknown_skeleton(stream, model, generic_hand_model, n_calibration_samples, outs[i], override)
        files.append(outs[i])
        if e is not None:
            errors.append(e)
    return (np.concatenate(errors) if errors else np.zeros(0)), files


def main():
    import argparse
    from .lib.common.hand import load_generic_hand_model
    from .lib.models.model_loader import load_pretrained_model
    ap = argparse.ArgumentParser(description="UmeTrack raw-data evaluation, one process per GPU "
                                             "(python -m torch.distributed.run --nproc-per-node N -m umetrack_amd.eval ...)")
    ap.add_argument("--protocol", choices=("known", "unknown"), default="known")
    ap.add_argument("--input", required=True, help="e.g. UmeTrack_data/raw_data/real")
    ap.add_argument("--output", required=True, help="e.g. tmp/eval_results_known_skeleton/real")
    ap.add_argument("--weights", required=True)
    ap.add_argument("--calibration-samples", type=int, default=30)
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    rank, world, local = udist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("umetrack_amd has no CPU path: an MI355X is required")
    torch.cuda.set_device(local)
    model = load_pretrained_model(args.weights).eval().to(torch.device("cuda", local))
    errs, files = run_eval(args.input, args.output, model, args.protocol, load_generic_hand_model(),
                           args.calibration_samples, rank=rank, world=world)
    if errs.size:
        logger.info("rank %d mean error: %s", rank, errs.mean())
    m = aggregate_metrics(args.output, files, reduce_over_ranks=True, verbose=rank == 0)
    if rank == 0 and m is None:
        print("no valid frames")


if __name__ == "__main__":
    main()
