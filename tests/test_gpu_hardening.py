"""Boundary behaviour where the reference raises or propagates NaN and an asynchronous kernel cannot:
the UMT_STATUS_* word, the read-all-before-write order of the temporal step, duplicates in memory_idx,
depth_check=False in _warp_image, the bounded timing event pools, and INTEGRATION.md section 1's
``sys.modules["lib.*"]`` aliasing driven by a caller written against the reference's names only."""
import importlib
import sys

import numpy as np
import pytest
import torch

from oracle import geometry as og
from umetrack_amd import _native as nat
from umetrack_amd import arch, synthetic

pytestmark = pytest.mark.gpu


def _model(seed=5):
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    return model_from_state_dict(arch.make_synthetic_state(seed)).to("cuda")


def _inputs(bs, seed, dev="cuda"):
    """bs two-view samples with plausible cameras; returns (InputFrameData, sample_range, skeleton)."""
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputSkeletonData
    g = torch.Generator().manual_seed(seed)
    n = 2 * bs
    imgs = torch.rand(n, 96, 96, generator=g)
    K = torch.zeros(n, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = 180 + 60 * torch.rand(n, generator=g)
    K[:, 0, 2] = K[:, 1, 2] = 47.5
    K[:, 2, 2] = 1
    E = torch.eye(4).repeat(n, 1, 1)
    E[:, :3, :3] = torch.from_numpy(synthetic.random_rotations(np.random.default_rng(seed), n)).float()
    E[:, :3, 3] = torch.rand(n, 3, generator=g) * 0.3
    sr = torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2])
    hm = scaled_hand_model(load_generic_hand_model(), 0.001)
    skel = InputSkeletonData(hm.joint_rotation_axes.to(dev), hm.joint_rest_positions.to(dev))
    return InputFrameData(imgs.to(dev), K.to(dev), E.to(dev)), sr.to(dev), skel


def _desc(sr, midx, use, dev="cuda"):
    from umetrack_amd.lib.models.umetrack_model import InputFrameDesc
    bs = sr.shape[0]
    return InputFrameDesc(sr, torch.as_tensor(midx).to(dev), torch.as_tensor(use, dtype=torch.bool).to(dev),
                          torch.zeros(bs, dtype=torch.long, device=dev))


def test_singular_previous_extrinsics_are_reported_not_propagated():
    """use_memory on a reserved but never written row: torch.inverse raises in the reference (temporal.py:66).
    Here the sample runs as use_memory=False, the state stays finite and the status word says so."""
    m, ref = _model(), _model()
    fd, sr, skel = _inputs(3, 1)
    a = m.regress_pose_use_skeleton(fd, _desc(sr, [0, 1, 2], [True, False, True]), skel)
    b = ref.regress_pose_use_skeleton(fd, _desc(sr, [0, 1, 2], [False, False, False]), skel)
    assert torch.equal(a.joint_angles, b.joint_angles) and torch.equal(a.wrist_xfs, b.wrist_xfs)
    mem, prev = m.memory_state()
    assert torch.isfinite(mem).all() and torch.equal(prev, fd.extrinsics_xf[0::2])
    assert m.status(clear=False) == 1
    with pytest.raises(nat.UmtError, match="singular"):
        m.check_status()
    assert m.status() == 0                      # check_status cleared it
    # the next step on the same rows has regular previous extrinsics: memory is used, nothing is flagged
    m.regress_pose_use_skeleton(fd, _desc(sr, [0, 1, 2], [True, True, True]), skel)
    m.check_status()
    assert ref.status() == 0


def test_duplicate_memory_idx_reads_before_writes_and_last_sample_wins():
    fd, sr, skel = _inputs(3, 2)
    m, ref = _model(), _model()
    for mod in (m, ref):                        # same first step: rows 0..2 hold regular state
        mod.regress_pose_use_skeleton(fd, _desc(sr, [0, 1, 2], [False] * 3), skel)
    fd2, _, _ = _inputs(3, 3)
    # samples 0 and 2 both name row 0: both must see the step-1 state of row 0, sample 2 writes it
    out = m.regress_pose_use_skeleton(fd2, _desc(sr, [0, 1, 0], [True] * 3), skel)
    mem, prev = m.memory_state()
    # reference run: sample 2 alone on row 0 (same previous state), sample 0 alone on a copy
    from umetrack_amd.lib.models.umetrack_model import InputFrameData
    pick = lambda fdx, rows: InputFrameData(fdx.left_images[rows], fdx.intrinsics[rows], fdx.extrinsics_xf[rows])  # noqa: E731
    one = torch.tensor([[0, 2]], device="cuda")
    o2 = ref.regress_pose_use_skeleton(pick(fd2, [4, 5]), _desc(one, [0], [True]), skel, required_rows=3)
    rmem, rprev = ref.memory_state()
    assert torch.equal(out.joint_angles[2], o2.joint_angles[0]) and torch.equal(out.wrist_xfs[2], o2.wrist_xfs[0])
    assert torch.equal(mem[0], rmem[0]) and torch.equal(prev[0], rprev[0])          # the last sample's state
    assert torch.equal(prev[0], fd2.extrinsics_xf[4])
    ref2 = _model()
    ref2.regress_pose_use_skeleton(fd, _desc(sr, [0, 1, 2], [False] * 3), skel)
    o0 = ref2.regress_pose_use_skeleton(pick(fd2, [0, 1]), _desc(one, [0], [True]), skel, required_rows=3)
    assert torch.equal(out.joint_angles[0], o0.joint_angles[0])                     # sample 0 read the OLD row 0
    m.check_status()


def test_out_of_range_indices_and_nan_inputs_are_flagged():
    m = _model()
    fd, sr, skel = _inputs(2, 4)
    # memory_idx beyond the reserved rows: runs without memory, flagged
    out = m.regress_pose_use_skeleton(fd, _desc(sr, [0, 7], [False, False]), skel, required_rows=2)
    assert torch.isfinite(out.joint_angles).all()
    assert m.status() == 2
    # a NaN pixel must not come out as a plausible pose (torch's ReLU / MaxPool propagate NaN)
    fd.left_images[2, 40, 40] = float("nan")
    out = m.regress_pose_use_skeleton(fd, _desc(sr, [0, 1], [False, False]), skel)
    assert torch.isfinite(out.joint_angles[0]).all() and torch.isnan(out.joint_angles[1]).all()
    assert torch.isnan(out.wrist_xfs[1]).all() and torch.isnan(out.landmark_uncertainty_sigmas[1]).all()
    # the reference returns such a NaN row and carries on: the status bit is set, check_status() reports it
    # without raising unless asked to
    assert m.status(clear=False) == nat.STATUS_NONFINITE_OUTPUT
    with pytest.raises(nat.UmtError, match="not finite"):
        m.check_status(raise_nonfinite=True)
    m.regress_pose_use_skeleton(fd, _desc(sr, [0, 1], [False, False]), skel)
    assert m.check_status() == nat.STATUS_NONFINITE_OUTPUT and m.status() == 0
    # malformed sample_range
    bad = sr.clone()
    bad[1, 1] = 9
    out = m.regress_pose_use_skeleton(fd, _desc(bad, [0, 1], [False, False]), skel)
    assert m.status() & 4


def test_fk_bad_bone_index():
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.common.hand_skinning import skin_landmarks
    hm = load_generic_hand_model()
    ja = torch.zeros(2, 22, device="cuda")
    wr = torch.eye(4, device="cuda").repeat(2, 1, 1)
    idx = hm.landmark_rest_bone_indices.clone()
    w = hm.landmark_rest_bone_weights
    lm, k = [int(v) for v in torch.nonzero(w != 0)[5]]
    idx[lm, k] = 17
    with pytest.raises(IndexError, match="out of bounds"):                      # host-resident model: checked before launch
        skin_landmarks(hm._replace(landmark_rest_bone_indices=idx), ja, wr)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = skin_landmarks(hm._replace(landmark_rest_bone_indices=idx.cuda()), ja, wr, status=status)
    good = skin_landmarks(hm, ja, wr)
    assert int(status.item()) == 16
    assert torch.isnan(out[:, lm]).all()
    keep = [i for i in range(21) if i != lm]
    assert torch.equal(out[:, keep], good[:, keep])
    # an out-of-range index with weight 0 is never used (hand_skinning.py:92-94)
    lm0, k0 = [int(v) for v in torch.nonzero(w == 0)[0]]
    idx2 = hm.landmark_rest_bone_indices.clone()
    idx2[lm0, k0] = 99
    status.zero_()
    out2 = skin_landmarks(hm._replace(landmark_rest_bone_indices=idx2.cuda()), ja, wr, status=status)
    assert int(status.item()) == 0 and torch.equal(out2, good)


def test_warp_image_without_depth_check_matches_oracle_maps():
    from umetrack_amd.lib.common import camera
    from umetrack_amd.lib.tracker.tracker import _warp_image
    rng = np.random.default_rng(31)
    intr, coeffs, c2w = synthetic.make_rig(rng)
    src = camera.Fisheye62CameraModel(synthetic.SRC_W, synthetic.SRC_H, tuple(intr[0, :2]), tuple(intr[0, 2:]),
                                      list(coeffs[0]), c2w[0])
    # a crop camera turned far enough from the source axis that part of the crop looks behind the source camera
    R = c2w[0].copy()
    ang = np.deg2rad(80.0)
    rot = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    R[:3, :3] = R[:3, :3] @ rot
    dst = camera.PinholePlaneCameraModel(96, 96, (40.0, 40.0), (47.5, 47.5), [], R)
    img = rng.integers(0, 256, size=(synthetic.SRC_H, synthetic.SRC_W)).astype(np.uint8)
    osrc = og.Fisheye62(src.width, src.height, src.f, src.c, list(src.distort), src.camera_to_world_xf)
    odst = og.Pinhole(96, 96, dst.f, dst.c, dst.camera_to_world_xf)
    seen_behind = False
    for depth_check in (True, False):
        mx, my = og.fisheye_warp_maps(osrc, odst, depth_check=depth_check)
        want = og.remap_linear_u8(img, mx, my)
        got = _warp_image(src, dst, img, depth_check=depth_check)
        assert got.dtype == np.uint8 and got.shape == (96, 96)
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.005, (depth_check, diff.max(), (diff > 0).mean())
        if not depth_check:
            mx1, _ = og.fisheye_warp_maps(osrc, odst, depth_check=True)
            seen_behind = bool(((mx1 == -1) & (mx != -1)).any())
    assert seen_behind, "the test camera never looks behind the source camera: depth_check is not exercised"
    with pytest.raises(NotImplementedError):
        _warp_image(src, dst, img, interpolation=0)


def test_timing_event_pools_are_bounded():
    m = _model()
    fd, sr, skel = _inputs(1, 6)
    d = _desc(sr, [0], [False])
    m.timing_enable(True)
    for _ in range(150):                        # more than two turns of the 64-slot ring
        m.regress_pose_use_skeleton(fd, d, skel, required_rows=1)
    b, f, calls = m.timing_read()
    assert calls == 150 and 0 < b <= f
    m.timing_enable(False)
    m.layer_timing_enable(True)
    for _ in range(50):                         # 50 passes x 27 marks > the 1024-mark cap: folded between passes
        m.regress_pose_use_skeleton(fd, d, skel, required_rows=1)
    rows = m.layer_timing_read()
    m.layer_timing_enable(False)
    assert len(rows) >= 20 and all(r[3] == 50 for r in rows) and all(r[2] > 0 for r in rows)


def test_reference_module_names_resolve_to_this_package(tmp_path):
    """alias umetrack_amd.lib.* as lib.*, then run a caller that only knows the
    reference's module paths: load -> HandTracker -> track_frame -> landmarks_from_hand_pose -> result file."""
    names = ("models", "models.model_loader", "models.umetrack_model", "models.regressor", "tracker", "tracker.tracker",
             "tracker.perspective_crop", "tracker.tracking_result", "common", "common.camera", "common.crop",
             "common.affine", "common.hand", "common.hand_skinning", "common.metric_utils", "data_utils",
             "data_utils.bundles")
    saved = {k: sys.modules.get(k) for k in ["lib"] + ["lib." + n for n in names]}
    try:
        sys.modules["lib"] = importlib.import_module("umetrack_amd.lib")
        for n in names:
            sys.modules["lib." + n] = importlib.import_module("umetrack_amd.lib." + n)
        # ---- from here on: reference names only (the body of run_eval_known_skeleton._track_sequence) ----
        import pickle
        import lib.common.camera as camera
        from lib.common.hand import NUM_HANDS, NUM_LANDMARKS_PER_HAND
        from lib.models.model_loader import load_pretrained_model
        from lib.tracker.perspective_crop import landmarks_from_hand_pose
        from lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
        from lib.tracker.tracking_result import SingleHandPose
        from lib.data_utils import bundles
        assert HandTracker.__module__ == "umetrack_amd.lib.tracker.tracker"
        weights = tmp_path / "pretrained_weights.torch"
        torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in arch.make_synthetic_state(5).items()}, weights)
        model = load_pretrained_model(str(weights))
        model.eval()
        tracker = HandTracker(model, HandTrackerOpts())
        rng = np.random.default_rng(9)
        intr, coeffs, c2w = synthetic.make_rig(rng)
        cams = [camera.Fisheye62CameraModel(synthetic.SRC_W, synthetic.SRC_H, tuple(intr[v, :2]), tuple(intr[v, 2:]),
                                            list(coeffs[v]), c2w[v]) for v in range(4)]
        from umetrack_amd.lib.common.hand import load_generic_hand_model      # the recording's hand model (label JSON)
        hand_model = load_generic_hand_model()
        n_frames = 3
        tracked = np.zeros([NUM_HANDS, n_frames, NUM_LANDMARKS_PER_HAND, 3])
        valid = np.zeros([NUM_HANDS, n_frames], dtype=bool)
        for f in range(n_frames):
            frame = rng.integers(0, 256, size=(synthetic.SRC_H, synthetic.SRC_W * 4)).astype(np.uint8)
            mv = frame.reshape(synthetic.SRC_H, 4, -1)
            views = [ViewData(image=mv[:, v, :], camera=cams[v], camera_angle=0.0) for v in range(4)]
            W = np.eye(4, dtype=np.float32)
            W[:3, 3] = (5.0 * f, -10.0, 430.0)
            gt = {0: SingleHandPose(joint_angles=np.zeros(22, np.float32), wrist_xform=W)}
            crop_cameras = tracker.gen_crop_cameras(cams, [0.0] * 4, hand_model, gt, min_num_crops=1)
            res = tracker.track_frame(InputFrame(views=views), hand_model, crop_cameras)
            for hand_idx in res.hand_poses:
                tracked[hand_idx, f] = landmarks_from_hand_pose(hand_model, res.hand_poses[hand_idx], hand_idx)
                valid[hand_idx, f] = True
        assert valid[0].all() and not valid[1].any() and np.isfinite(tracked).all()
        assert tracker._valid_tracking_history.tolist() == [True, False]
        out_path = tmp_path / "rec.npy"
        with open(out_path, "wb") as fp:
            pickle.dump({"tracked_keypoints": tracked, "gt_keypoints": tracked.copy(), "valid_tracking": valid}, fp)
        assert bundles.to_device({"a": torch.zeros(2)}, "cpu")["a"].device.type == "cpu"
        # ---- and this package reads the file back (load_eval.aggregate_metrics) ----
        from umetrack_amd import eval as ev
        agg = ev.aggregate_metrics(str(tmp_path), verbose=False)
        assert agg.n_valid == 3 and agg.n_total == 6 and agg.mpjpe == 0.0
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
