"""GPU crop-camera generation (rows A0/F1) against the host path, which follows the reference's
numpy code and is pinned to its goldens (tests/test_host_logic.py).  Tolerance: 1e-9 relative on
the float64 camera parameters (same formulas, different summation order and libm), exact view
selection."""
import numpy as np
import pytest
import torch

from umetrack_amd import synthetic
from umetrack_amd.lib.common import camera
from umetrack_amd.lib.common.hand import load_generic_hand_model
from umetrack_amd.lib.tracker import crop_cameras as cc
from umetrack_amd.lib.tracker import perspective_crop as pc
from umetrack_amd.lib.tracker import warp as gpu_warp
from umetrack_amd.lib.tracker.tracking_result import SingleHandPose

pytestmark = pytest.mark.gpu


def _rig(rng, wide):
    intr, coeffs, c2w = synthetic.make_rig(rng)
    if not wide:
        intr = intr.copy()
        intr[:, :2] = 420.0                      # narrow field of view: some hands fall outside some cameras
    cams = [camera.Fisheye62CameraModel(synthetic.SRC_W, synthetic.SRC_H, intr[v, :2], intr[v, 2:], list(coeffs[v]), c2w[v])
            for v in range(4)]
    packed = np.stack([gpu_warp.pack_fisheye(synthetic.SRC_W, synthetic.SRC_H, intr[v, :2], intr[v, 2:], coeffs[v], c2w[v])
                       for v in range(4)])
    return cams, packed


@pytest.mark.parametrize("wide,sort_index,npts", [(True, True, 63), (False, True, 63), (False, False, 42), (True, False, 21)])
def test_matches_host_path(wide, sort_index, npts):
    rng = np.random.default_rng(11 + npts)
    hm = load_generic_hand_model()
    cams, packed = _rig(rng, wide)
    angles = [0.0, 90.0, -90.0, 180.0]
    B = 48
    lim = hm.joint_limits.numpy().astype(np.float64)
    ja = rng.uniform(lim[:, 0], lim[:, 1], size=(B, 22)).astype(np.float32)
    ja[:, 20:] = 0
    wrist = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    wrist[:, :3, :3] = synthetic.random_rotations(rng, B)
    wrist[:, :3, 3] = np.stack([rng.uniform(-200, 200, B), rng.uniform(-150, 150, B), rng.uniform(300, 500, B)], -1)
    hand = rng.integers(0, 2, B)
    pts = cc.crop_points_batched(hm, torch.from_numpy(ja), torch.from_numpy(wrist), torch.from_numpy(hand), npts)
    got = cc.gen_crop_cameras_batched(pts, torch.from_numpy(packed), torch.tensor(angles), torch.from_numpy(hand),
                                      sort_camera_index=sort_index)
    nv, vi = got.n_views.cpu().numpy(), got.view_idx.cpu().numpy()
    dc, K, E, st = (t.cpu().numpy() for t in (got.dst_cams, got.intrinsics, got.extrinsics, got.status))
    seen_counts = set()
    for b in range(B):
        pose = SingleHandPose(joint_angles=ja[b], wrist_xform=wrist[b])
        try:
            want = pc.gen_crop_cameras_from_pose(cams, angles, hm, pose, int(hand[b]), npts, (96, 96), max_view_num=2,
                                                 sort_camera_index=sort_index)
            raised = False
        except ValueError:
            raised = True
        if raised:
            assert st[b, :nv[b]].any(), b
            continue
        seen_counts.add(len(want))
        assert nv[b] == len(want), (b, nv[b], list(want))
        assert list(vi[b, :nv[b]]) == list(want) and (vi[b, nv[b]:] == -1).all()
        assert not st[b].any()
        for k, (v, cam) in enumerate(want.items()):
            np.testing.assert_allclose(dc[b, k, :2], cam.f, rtol=1e-9)
            np.testing.assert_allclose(dc[b, k, 2:4], cam.c, rtol=0, atol=0)
            np.testing.assert_allclose(dc[b, k, 4:].reshape(4, 4), cam.camera_to_world_xf, rtol=0, atol=1e-8)
            np.testing.assert_allclose(K[b, k], cam.uv_to_window_matrix().astype(np.float32), rtol=1e-6)
            e = np.linalg.inv(cam.camera_to_world_xf)
            e[:3, 3] *= 0.001
            np.testing.assert_allclose(E[b, k], e.astype(np.float32), rtol=0, atol=2e-6)
    if not wide:
        assert len(seen_counts) > 1, "the narrow rig should produce hands with 0/1/2 views"


def test_forced_views_and_degenerate_points():
    rng = np.random.default_rng(5)
    hm = load_generic_hand_model()
    _, packed = _rig(rng, True)
    B = 6
    pts = torch.zeros(B, 63, 3)
    pts[:] = torch.tensor([0.0, 0.0, 400.0]) + torch.randn(B, 63, 3) * 30
    pts[3, 5] = torch.tensor([0.0, 0.0, -50.0])                   # one crop point behind every camera
    forced = torch.tensor([[0, 1], [2, 3], [1, -1], [0, 1], [3, 2], [-1, -1]], dtype=torch.int32)
    got = cc.gen_crop_cameras_batched(pts, torch.from_numpy(packed), torch.zeros(4), torch.zeros(B, dtype=torch.long),
                                      forced_views=forced)
    assert got.n_views.tolist() == [2, 2, 1, 2, 2, 0]
    assert got.view_idx.tolist() == [[0, 1], [2, 3], [1, -1], [0, 1], [3, 2], [-1, -1]]
    st = got.status.cpu().numpy()
    assert st[3].all() and not st[[0, 1, 4]].any()
    # synthetic.crop_cameras_batched (the bench's host setup) agrees with the kernel on the same views
    c2w = packed[:, 14:].reshape(4, 4, 4)
    v = np.array([0, 1, 2, 3])
    f, c, crop_c2w, ok = synthetic.crop_cameras_batched(c2w[v], np.zeros(4), pts[:2].numpy().repeat(2, 0),
                                                        np.zeros(4, bool))
    np.testing.assert_allclose(got.dst_cams.cpu().numpy()[:2, :, 0].reshape(-1), f, rtol=1e-9)
    np.testing.assert_allclose(got.dst_cams.cpu().numpy()[:2, :, 4:].reshape(4, 4, 4), crop_c2w, atol=1e-8)


def test_argument_checks():
    from umetrack_amd import _native as nat
    pts = torch.zeros(2, 63, 3, device="cuda")
    with pytest.raises(ValueError, match="cameras must be"):
        cc.gen_crop_cameras_batched(pts, torch.zeros(4, 29, dtype=torch.float64), torch.zeros(4), torch.zeros(2, dtype=torch.long))
    with pytest.raises(nat.UmtError, match="crop points"):
        cc.gen_crop_cameras_batched(torch.zeros(2, 70, 3), torch.zeros(4, 30, dtype=torch.float64), torch.zeros(4),
                                    torch.zeros(2, dtype=torch.long))
    with pytest.raises(nat.UmtError, match="max_views"):
        cc.gen_crop_cameras_batched(pts, torch.zeros(4, 30, dtype=torch.float64), torch.zeros(4),
                                    torch.zeros(2, dtype=torch.long), max_view_num=5)
