"""HandTracker end to end on the GPU (warp -> network -> result dicts) against the oracle pipeline,
plus lock-step recurrent recordings (temporal memory on) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import fk as ofk
from oracle import geometry as og
from oracle import model as om
from umetrack_amd import arch, synthetic

pytestmark = pytest.mark.gpu


def _rig_cameras(rng):
    from umetrack_amd.lib.common import camera
    intr, coeffs, c2w = synthetic.make_rig(rng)
    return [camera.Fisheye62CameraModel(synthetic.SRC_W, synthetic.SRC_H, tuple(intr[v, :2]), tuple(intr[v, 2:]),
                                        list(coeffs[v]), c2w[v]) for v in range(4)]


def test_tracker_frames_match_oracle(hand_model_np):
    from types import SimpleNamespace
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.tracker.perspective_crop import landmarks_from_hand_pose
    from umetrack_amd.lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    from umetrack_amd.lib.tracker.tracking_result import SingleHandPose
    rng = np.random.default_rng(21)
    state = arch.make_synthetic_state(5)
    hm = load_generic_hand_model()
    tracker = HandTracker(model_from_state_dict(state), HandTrackerOpts())
    oracle = om.OracleModel(state)
    cams = _rig_cameras(rng)
    lim = hand_model_np["joint_limits"]
    history = np.zeros(2, bool)
    skel = SimpleNamespace(joint_rotation_axes=torch.from_numpy(hand_model_np["joint_rotation_axes"]).float(),
                           joint_rest_positions=torch.from_numpy(hand_model_np["joint_rest_positions"]).float() * 0.001)
    for frame_idx in range(4):
        frame = rng.integers(0, 256, size=(synthetic.SRC_H, synthetic.SRC_W * 4)).astype(np.uint8)
        mv = frame.reshape(synthetic.SRC_H, 4, -1)                     # video_pose_data.py:123-125
        views = [ViewData(image=mv[:, v, :], camera=cams[v], camera_angle=0.0) for v in range(4)]
        gt = {}
        hands = (0, 1) if frame_idx != 2 else (1,)                     # left hand drops out once
        for h in hands:
            ja = rng.uniform(lim[:, 0], lim[:, 1]).astype(np.float32)
            ja[20:] = 0
            W = np.eye(4, dtype=np.float32)
            W[:3, :3] = synthetic.random_rotations(rng, 1)[0]
            W[:3, 3] = (rng.uniform(-60, 60), rng.uniform(-40, 40), rng.uniform(380, 480))
            gt[h] = SingleHandPose(joint_angles=ja, wrist_xform=W, hand_confidence=1.0)
        crop_cams = tracker.gen_crop_cameras(cams, [0.0] * 4, hm, gt, min_num_crops=1)
        assert set(crop_cams) == set(hands)
        res = tracker.track_frame(InputFrame(views=views), hm, crop_cams)
        # ---- oracle: same crop cameras, oracle warp + oracle network ----
        crops, K, E, ranges, hidx = [], [], [], [], []
        for h, per_hand in crop_cams.items():
            first = len(crops)
            for ci, cc in per_hand.items():
                src = og.Fisheye62(cams[ci].width, cams[ci].height, cams[ci].f, cams[ci].c, list(cams[ci].distort),
                                   cams[ci].camera_to_world_xf)
                dst = og.Pinhole(96, 96, cc.f, cc.c, cc.camera_to_world_xf)
                crops.append(og.warp_crop(src, dst, np.ascontiguousarray(mv[:, ci, :])))
                K.append(cc.uv_to_window_matrix())
                w2e = np.linalg.inv(cc.camera_to_world_xf)
                w2e[:3, 3] *= 0.001
                E.append(w2e)
            ranges.append((first, len(crops)))
            hidx.append(h)
        fd = SimpleNamespace(left_images=torch.from_numpy(np.stack(crops)), intrinsics=torch.from_numpy(np.stack(K)).float(),
                             extrinsics_xf=torch.from_numpy(np.stack(E)).float())
        desc = SimpleNamespace(sample_range=torch.tensor(ranges), memory_idx=torch.tensor(hidx),
                               use_memory=torch.from_numpy(history[hidx]), hand_idx=torch.tensor(hidx))
        with torch.no_grad():
            ref = oracle.regress_pose_use_skeleton(fd, desc, skel)
        history[:] = False
        history[hidx] = True
        assert set(res.hand_poses) == set(hands) and res.num_views == {h: len(crop_cams[h]) for h in hands}
        for row, h in enumerate(hidx):
            pose = res.hand_poses[h]
            wref = ref["wrist_xfs"][row].numpy().copy()
            wref[:3, 3] *= 1000.0
            # mm; the warp may differ from the oracle by rare one-grey-level flips -> small feature noise
            assert np.abs(pose.wrist_xform[:3, 3] - wref[:3, 3]).max() < 0.02, (frame_idx, h)
            assert np.abs(pose.joint_angles - ref["joint_angles"][row].numpy()).max() < 1e-4
            lm = landmarks_from_hand_pose(hm, pose, h)
            lm_ref = ofk.landmarks_from_pose(hand_model_np, ref["joint_angles"][row].numpy(), wref, h, np.float64)
            assert np.abs(lm - lm_ref).max() < 0.03
    # a frame without hands resets the history and returns an empty result
    empty = tracker.track_frame(InputFrame(views=views), hm, {})
    assert empty.hand_poses == {} and not tracker._valid_tracking_history.any()


def test_track_frame_one_host_sync_and_cached_landmarks(monkeypatch):
    """a frame costs ONE host synchronisation.  After a warm-up frame (allocations, memory rows)
    track_frame runs with torch's sync debug mode set to "warn" - every pageable copy, .item(), .cpu() and explicit
    synchronisation warns - and both the warnings and the explicit synchronisations are counted.  The landmarks the result carries equal those of the
    generic landmarks_from_hand_pose round trip."""
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.tracker.perspective_crop import landmarks_from_hand_pose
    from umetrack_amd.lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    from umetrack_amd.lib.tracker.tracking_result import SingleHandPose
    rng = np.random.default_rng(23)
    hm = load_generic_hand_model()
    tracker = HandTracker(model_from_state_dict(arch.make_synthetic_state(5)), HandTrackerOpts())
    cams = _rig_cameras(rng)
    frames, crops = [], []
    for _ in range(3):
        frame = rng.integers(0, 256, size=(synthetic.SRC_H, synthetic.SRC_W * 4)).astype(np.uint8)
        mv = frame.reshape(synthetic.SRC_H, 4, -1)
        frames.append(InputFrame(views=[ViewData(image=mv[:, v, :], camera=cams[v], camera_angle=0.0) for v in range(4)]))
        gt = {}
        for h in (0, 1):
            W = np.eye(4, dtype=np.float32)
            W[:3, :3] = synthetic.random_rotations(rng, 1)[0]
            W[:3, 3] = (rng.uniform(-60, 60), rng.uniform(-40, 40), rng.uniform(380, 480))
            gt[h] = SingleHandPose(joint_angles=np.zeros(22, np.float32), wrist_xform=W, hand_confidence=1.0)
        crops.append(tracker.gen_crop_cameras(cams, [0.0] * 4, hm, gt, min_num_crops=1))
    tracker.track_frame(frames[0], hm, crops[0])               # warm-up
    torch.cuda.synchronize()
    count = {"n": 0}
    real_stream_sync, real_sync = torch.cuda.Stream.synchronize, torch.cuda.synchronize

    def stream_sync(self):
        count["n"] += 1
        return real_stream_sync(self)

    def dev_sync(*a, **k):
        count["n"] += 1
        return real_sync(*a, **k)

    monkeypatch.setattr(torch.cuda.Stream, "synchronize", stream_sync)
    monkeypatch.setattr(torch.cuda, "synchronize", dev_sync)
    import warnings
    torch.cuda.set_sync_debug_mode("warn")       # every synchronising torch call warns: implicit ones and the explicit one
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            results = [tracker.track_frame(frames[i], hm, crops[i]) for i in (1, 2)]
    finally:
        torch.cuda.set_sync_debug_mode("default")
        monkeypatch.undo()
    sync_warnings = [w for w in caught if "synchronizing" in str(w.message)]
    assert len(sync_warnings) == 2, [str(w.message) for w in caught]      # pageable copies, .item(), .cpu() would add to it
    assert count["n"] == 2, count                              # one per frame
    for res in results:
        assert set(res.hand_poses) == {0, 1}
        for h, pose in res.hand_poses.items():
            assert isinstance(pose, SingleHandPose) and pose.landmarks_mm is not None
            fast = landmarks_from_hand_pose(hm, pose, h)
            assert fast is pose.landmarks_mm
            slow = landmarks_from_hand_pose(hm, SingleHandPose(*pose), h)      # plain tuple: the FK round trip
            assert np.abs(fast - slow).max() < 1e-3 and np.isfinite(fast).all()
            other = landmarks_from_hand_pose(load_generic_hand_model(), pose, h)   # another model object: no cache hit
            assert other is not pose.landmarks_mm


def test_tracker_staging_block_across_frame_shapes():
    """The tracker keeps ONE pinned staging block whose pixel part only grows.  A tracker that has seen 640x480 views and
    is then given 320x240 views (other cameras, other pitch, fewer bytes), and then large ones again, must return exactly
    what a fresh tracker returns for each of those frames (no history: first-frame results are comparable)."""
    from umetrack_amd.lib.common import camera
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    from umetrack_amd.lib.tracker.tracking_result import SingleHandPose
    rng = np.random.default_rng(29)
    hm = load_generic_hand_model()
    state = arch.make_synthetic_state(5)
    big = _rig_cameras(rng)
    small = [camera.Fisheye62CameraModel(c.width // 2, c.height // 2, tuple(np.asarray(c.f) / 2), tuple(np.asarray(c.c) / 2),
                                         list(c.distort), c.camera_to_world_xf) for c in big]
    gt = {}
    for h in (0, 1):
        W = np.eye(4, dtype=np.float32)
        W[:3, :3] = synthetic.random_rotations(rng, 1)[0]
        W[:3, 3] = (rng.uniform(-60, 60), rng.uniform(-40, 40), rng.uniform(380, 480))
        gt[h] = SingleHandPose(joint_angles=np.zeros(22, np.float32), wrist_xform=W, hand_confidence=1.0)

    def frame_for(cams):
        img = rng.integers(0, 256, size=(cams[0].height, cams[0].width * 4)).astype(np.uint8)
        mv = img.reshape(cams[0].height, 4, -1)
        return InputFrame(views=[ViewData(image=mv[:, v, :], camera=cams[v], camera_angle=0.0) for v in range(4)])

    seq = [(big, frame_for(big)), (small, frame_for(small)), (big, frame_for(big)), (small, frame_for(small))]
    walker = HandTracker(model_from_state_dict(state), HandTrackerOpts())
    for cams, frame in seq:
        crop_cams = walker.gen_crop_cameras(cams, [0.0] * 4, hm, gt, min_num_crops=1)
        walker.reset_history(clear_model_memory=True)
        got = walker.track_frame(frame, hm, crop_cams)
        fresh = HandTracker(model_from_state_dict(state), HandTrackerOpts())
        want = fresh.track_frame(frame, hm, crop_cams)
        assert set(got.hand_poses) == set(want.hand_poses) == {0, 1}
        for h in (0, 1):
            assert np.array_equal(got.hand_poses[h].joint_angles, want.hand_poses[h].joint_angles)
            assert np.array_equal(got.hand_poses[h].wrist_xform, want.hand_poses[h].wrist_xform)
            assert np.array_equal(got.hand_poses[h].landmarks_mm, want.hand_poses[h].landmarks_mm)
            assert np.isfinite(got.hand_poses[h].landmarks_mm).all()


def test_scale_calibration_head_through_tracker(hand_model_np):
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    from umetrack_amd.lib.tracker.tracking_result import SingleHandPose
    rng = np.random.default_rng(22)
    hm = load_generic_hand_model()
    tracker = HandTracker(model_from_state_dict(arch.make_synthetic_state(5)), HandTrackerOpts())
    cams = _rig_cameras(rng)
    frame = rng.integers(0, 256, size=(synthetic.SRC_H, synthetic.SRC_W * 4)).astype(np.uint8)
    mv = frame.reshape(synthetic.SRC_H, 4, -1)
    views = [ViewData(image=mv[:, v, :], camera=cams[v], camera_angle=0.0) for v in range(4)]
    W = np.eye(4, dtype=np.float32)
    W[:3, 3] = (10, -5, 420)
    gt = {0: SingleHandPose(joint_angles=np.zeros(22, np.float32), wrist_xform=W)}
    crop_cams = tracker.gen_crop_cameras(cams, [0.0] * 4, hm, gt, min_num_crops=2)
    res = tracker.track_frame_and_calibrate_scale(InputFrame(views=views), crop_cams)
    assert set(res.predicted_scales) == {0} and res.predicted_scales[0] > 0
    # single-view samples are rejected by the scale head exactly like the reference asserts
    one = {0: dict(list(crop_cams[0].items())[:1])}
    with pytest.raises(AssertionError, match="single-view"):
        tracker.track_frame_and_calibrate_scale(InputFrame(views=views), one)


def test_recurrent_lockstep_recordings_match_oracle():
    """64 recordings in lock-step, 6 frames, memory on after the first frame (section 8d, second mode)."""
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    state = arch.make_synthetic_state(9)
    gm = model_from_state_dict(state).to("cuda")
    oracle = om.OracleModel(state)
    rng = np.random.default_rng(31)
    n_rec, bs = 64, 128
    hm = scaled_hand_model(load_generic_hand_model(), 0.001)
    skel = InputSkeletonData(hm.joint_rotation_axes, hm.joint_rest_positions)
    base_E = np.tile(np.eye(4, dtype=np.float32), (2 * bs, 1, 1))
    from scipy.spatial.transform import Rotation
    base_E[:, :3, :3] = Rotation.from_rotvec(rng.normal(size=(2 * bs, 3)) * 0.4).as_matrix()
    base_E[:, :3, 3] = rng.normal(size=(2 * bs, 3)) * 0.2
    worst = 0.0
    for t in range(6):
        img = torch.from_numpy(rng.integers(0, 256, size=(2 * bs, 96, 96)).astype(np.float32) / np.float32(255))
        K = torch.zeros(2 * bs, 3, 3)
        K[:, 0, 0] = K[:, 1, 1] = torch.from_numpy(rng.uniform(150, 400, 2 * bs).astype(np.float32))
        K[:, 0, 2] = K[:, 1, 2] = 47.5
        K[:, 2, 2] = 1
        E = base_E.copy()
        E[:, :3, 3] += rng.normal(size=(2 * bs, 3)).astype(np.float32) * 0.01      # the rig moves a little every frame
        use = torch.full((bs,), t > 0)
        if t == 3:
            use[::7] = False                                                        # some hands re-acquired
        fd = InputFrameData(img, K, torch.from_numpy(E))
        desc = InputFrameDesc(torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(bs), use,
                              torch.arange(bs) % 2)
        got = gm.regress_pose_use_skeleton(fd, desc, skel)
        with torch.no_grad():
            ref = oracle.regress_pose_use_skeleton(fd, desc, skel)
        d = np.abs(got.wrist_xfs.cpu().numpy()[:, :3, 3] - ref["wrist_xfs"].numpy()[:, :3, 3]).max()
        worst = max(worst, float(d))
        assert d < 1e-5, (t, d)                                                     # 0.01 mm
    mem, _ = gm.memory_state()
    assert mem.shape[0] == bs and n_rec * 2 == bs
    rel = float((mem.cpu() - oracle.state.mem).abs().max() / oracle.state.mem.abs().max())
    assert rel < 2e-4, rel


def test_bulk_runner_equals_frame_by_frame(hand_model_np):
    """Backbone in bulk + lock-step temporal steps == one regress call per frame, and both match the oracle."""
    from umetrack_amd import bulk
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.common.hand_skinning import skin_landmarks
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    from umetrack_amd.lib.tracker import warp as gpu_warp
    dev = torch.device("cuda", 0)
    hm = load_generic_hand_model()

    def fk(ja, wr, hidx):
        return skin_landmarks(hm, torch.from_numpy(ja), torch.from_numpy(wr), hand_idx=torch.from_numpy(hidx),
                              mirror_right=True).cpu().numpy()
    n_rec, T = 3, 5
    wl = synthetic.make_workload(n_rec * T, T, 7, hm, fk)
    state = arch.make_synthetic_state(13)
    model = model_from_state_dict(state).to(dev)
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)  # noqa: E731
    images = torch.randint(0, 256, (wl.image_bytes,), dtype=torch.uint8, device=dev)
    off, pit = d(wl.src_offset, torch.int64), d(wl.src_pitch, torch.int32)
    sc, dc = d(wl.src_cams, torch.float64), d(wl.dst_cams, torch.float64)
    K, E, hidx = d(wl.intrinsics, torch.float32), d(wl.extrinsics, torch.float32), d(wl.hand_idx, torch.int64)
    hm_m = scaled_hand_model(hm, 0.001)
    skel = InputSkeletonData(hm_m.joint_rotation_axes.to(dev), hm_m.joint_rest_positions.to(dev))
    valid = torch.ones(n_rec * 2, T, dtype=torch.bool, device=dev)
    valid[1, 2] = False                                             # one hand loses tracking for a frame
    runner = bulk.BulkSequenceRunner(model, n_rec, T, chunk_samples=8)   # chunking that does not divide S
    runner.features(images, off, pit, sc, dc, K, E)
    got = runner.run(E, hidx, skel, valid)
    # the one-launch sequence kernel (default) and the stepwise loop give the same bits, state included
    mem_seq, prev_seq = model.memory_state()
    step = runner.run(E, hidx, skel, valid, stepwise=True)
    mem_step, prev_step = model.memory_state()
    keep_all = valid.reshape(-1)
    assert torch.equal(got.joint_angles[keep_all], step.joint_angles[keep_all])
    assert torch.equal(got.wrist_xfs[keep_all], step.wrist_xfs[keep_all])
    assert torch.equal(got.landmark_uncertainty_sigmas[keep_all], step.landmark_uncertainty_sigmas[keep_all])
    assert torch.equal(mem_seq, mem_step) and torch.equal(prev_seq, prev_step)
    model.check_status()
    # frame-by-frame on a second model instance + the oracle
    model2 = model_from_state_dict(state).to(dev)
    oracle = om.OracleModel(state)
    R = n_rec * 2
    crops = gpu_warp.warp_crops(images, off, pit, sc, dc)
    prev_ok = torch.zeros(R, dtype=torch.bool)
    for t in range(T):
        idx = torch.arange(R) * T + t
        cidx = torch.stack([2 * idx, 2 * idx + 1], 1).reshape(-1).to(dev)
        fd = InputFrameData(crops[cidx], K[cidx], E[cidx])
        desc = InputFrameDesc(torch.arange(R)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(R), prev_ok,
                              hidx[idx.to(dev)])
        ref = model2.regress_pose_use_skeleton(fd, desc, skel)
        with torch.no_grad():
            orc = oracle.regress_pose_use_skeleton(fd, desc, skel)
        keep = valid[:, t].cpu()
        a = got.wrist_xfs[idx.to(dev)].cpu()[keep]
        assert torch.equal(a, ref.wrist_xfs.cpu()[keep]), t                       # identical arithmetic
        assert torch.equal(got.joint_angles[idx.to(dev)].cpu()[keep], ref.joint_angles.cpu()[keep])
        assert (a[:, :3, 3] - orc["wrist_xfs"][keep][:, :3, 3]).abs().max() < 1e-5
        prev_ok = valid[:, t].cpu()


@pytest.mark.parametrize("one_kernel", [False, True])
@pytest.mark.parametrize("head_name,per_row_skel", [("known", False), ("known", True), ("scale", False)])
def test_head_sequence_equals_stepwise_calls(monkeypatch, one_kernel, head_name, per_row_skel):
    """umt_head_sequence (temporal module per memory row in one launch + the regressor batched over all T x R rows; with
    UMT_SEQ_ONEKERNEL the regressor inside that launch, step by step) == T calls of temporal_step + regress_decode,
    bit for bit: both heads, shared and per-row skeletons, memory switched off for some rows mid-sequence."""
    from umetrack_amd import _native as nat
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputSkeletonData
    if one_kernel:
        monkeypatch.setenv("UMT_SEQ_ONEKERNEL", "1")
    else:
        monkeypatch.delenv("UMT_SEQ_ONEKERNEL", raising=False)
    dev = torch.device("cuda", 0)
    model = model_from_state_dict(arch.make_synthetic_state(21)).to(dev)       # the switch is read per model handle
    head = nat.HEAD_KNOWN_SKELETON if head_name == "known" else nat.HEAD_PREDICT_SCALE
    T, R = 5, 7
    g = torch.Generator().manual_seed(5)
    fused_t = torch.randn(T, R, 72, 6, 6, generator=g).to(dev)
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    cam0 = np.tile(np.eye(4, dtype=np.float32), (T, R, 1, 1))
    cam0[..., :3, :3] = Rotation.from_rotvec(rng.normal(size=(T * R, 3)) * 0.3).as_matrix().reshape(T, R, 3, 3)
    cam0[..., :3, 3] = rng.normal(size=(T, R, 3)) * 0.1
    cam0_t = torch.from_numpy(cam0).to(dev)
    use_t = torch.ones(T, R, dtype=torch.bool, device=dev)
    use_t[0] = False
    use_t[3, ::3] = False
    hand_rows = (torch.arange(R) % 2).to(dev)
    skel = None
    if head == nat.HEAD_KNOWN_SKELETON:
        hm = load_generic_hand_model()
        if per_row_skel:
            hms = [scaled_hand_model(hm, 0.001 * (0.9 + 0.03 * r)) for r in range(R)]
            skel = InputSkeletonData(torch.stack([h.joint_rotation_axes for h in hms]).to(dev),
                                     torch.stack([h.joint_rest_positions for h in hms]).to(dev))
        else:
            hm_m = scaled_hand_model(hm, 0.001)
            skel = InputSkeletonData(hm_m.joint_rotation_axes.to(dev), hm_m.joint_rest_positions.to(dev))
    model.reset_memory()
    seq = model.head_sequence(head, fused_t, cam0_t, use_t, hand_rows.repeat(T, 1), skel)
    mem_seq, prev_seq = [x.clone() for x in model.memory_state()]
    model.reset_memory()
    midx = torch.arange(R, device=dev)
    for t in range(T):
        temporal = model.temporal_step(fused_t[t], cam0_t[t], midx, use_t[t], required_rows=R)
        out, _ = model.regress_decode(head, temporal, cam0_t[t], hand_rows, skel)
        assert torch.equal(seq.joint_angles[t], out.joint_angles), t
        assert torch.equal(seq.wrist_xfs[t], out.wrist_xfs), t
        assert torch.equal(seq.landmark_uncertainty_sigmas[t], out.landmark_uncertainty_sigmas), t
        if head == nat.HEAD_PREDICT_SCALE:
            assert torch.equal(seq.skel_scales[t], out.skel_scales), t
    mem_step, prev_step = model.memory_state()
    assert torch.equal(mem_seq, mem_step) and torch.equal(prev_seq, prev_step)
    model.check_status()


def test_head_sequence_regressor_batches_cover_long_calls():
    """umt_head_sequence runs its regressor over the T x R hand-frames in batches of at most 8192: a call with 9000 of them
    (two batches, the second one short) must still equal the per-step calls bit for bit."""
    from umetrack_amd import _native as nat
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputSkeletonData
    dev = torch.device("cuda", 0)
    model = model_from_state_dict(arch.make_synthetic_state(22)).to(dev)
    T, R = 3, 3000
    g = torch.Generator(device=dev).manual_seed(6)
    fused_t = torch.randn(T, R, 72, 6, 6, generator=g, device=dev)
    cam0_t = torch.eye(4, device=dev).repeat(T, R, 1, 1)
    cam0_t[..., :3, 3] = torch.randn(T, R, 3, generator=g, device=dev) * 0.1
    use_t = torch.ones(T, R, dtype=torch.bool, device=dev)
    use_t[0] = False
    hand_rows = (torch.arange(R, device=dev) % 2)
    hm_m = scaled_hand_model(load_generic_hand_model(), 0.001)
    skel = InputSkeletonData(hm_m.joint_rotation_axes.to(dev), hm_m.joint_rest_positions.to(dev))
    model.reset_memory()
    seq = model.head_sequence(nat.HEAD_KNOWN_SKELETON, fused_t, cam0_t, use_t, hand_rows.repeat(T, 1), skel)
    mem_seq, _ = [x.clone() for x in model.memory_state()]
    model.reset_memory()
    midx = torch.arange(R, device=dev)
    for t in range(T):
        temporal = model.temporal_step(fused_t[t], cam0_t[t], midx, use_t[t], required_rows=R)
        out, _ = model.regress_decode(nat.HEAD_KNOWN_SKELETON, temporal, cam0_t[t], hand_rows, skel)
        assert torch.equal(seq.joint_angles[t], out.joint_angles), t
        assert torch.equal(seq.wrist_xfs[t], out.wrist_xfs), t
        assert torch.equal(seq.landmark_uncertainty_sigmas[t], out.landmark_uncertainty_sigmas), t
    assert torch.equal(mem_seq, model.memory_state()[0])
    model.check_status()
