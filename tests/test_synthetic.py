"""The batched host geometry used by bench/smoke equals the per-hand reference-style path."""
import numpy as np

from oracle import fk as ofk
from umetrack_amd import synthetic
from umetrack_amd.lib.common import camera, crop
from umetrack_amd.lib.common.hand import load_generic_hand_model


def _fk(hand_model_np):
    def fn(ja, wr, hidx):
        w = wr.astype(np.float64).copy()
        w[hidx == 1, :, 0] *= -1
        return ofk.skin_landmarks(hand_model_np, ja, w, np.float32)
    return fn


def test_workload_crop_cameras_equal_scalar_path(hand_model_np):
    hm = load_generic_hand_model()
    wl = synthetic.make_workload(8, 4, 1, hm, _fk(hand_model_np))
    S = wl.joint_angles.shape[0]
    assert S == 16 and wl.src_cams.shape == (32, 30) and wl.dst_cams.shape == (32, 20)
    assert (wl.hand_idx.reshape(2, 2, 4)[:, 0] == 0).all() and (wl.hand_idx.reshape(2, 2, 4)[:, 1] == 1).all()
    lim = hand_model_np["joint_limits"]
    neutral = (lim[:, 0] * 0.5 + lim[:, 1] * 0.5).astype(np.float32)
    fk = _fk(hand_model_np)
    for s in (0, 5, 11, 15):
        poses = np.stack([wl.joint_angles[s], neutral, np.zeros(22, np.float32)])
        pts = fk(poses, np.repeat(wl.wrist[s:s + 1], 3, 0), np.repeat(wl.hand_idx[s:s + 1], 3)).reshape(63, 3)
        for v in range(2):
            k = 2 * s + v
            sc = wl.src_cams[k]
            src = camera.Fisheye62CameraModel(int(sc[0]), int(sc[1]), (sc[2], sc[3]), (sc[4], sc[5]), list(sc[6:14]),
                                              sc[14:].reshape(4, 4))
            ref = crop.gen_crop_parameters_from_points(src, pts, (96, 96), bool(wl.hand_idx[s] == 1), 0, 0.95)
            np.testing.assert_allclose(wl.dst_cams[k, :2], ref.f, rtol=1e-10)
            np.testing.assert_allclose(wl.dst_cams[k, 4:].reshape(4, 4), ref.camera_to_world_xf, atol=1e-8)
            w2e = np.linalg.inv(ref.camera_to_world_xf)
            w2e[:3, 3] *= 0.001
            np.testing.assert_allclose(wl.extrinsics[k], w2e, atol=1e-5)
            assert wl.src_offset[k] == wl.frame_of[s] * 480 * 2560 + wl.views[s, v] * 640
