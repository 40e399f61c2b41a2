"""Raw-data recordings on disk: label JSON + frames -> (InputFrame,
gt_tracking), the logic of lib/tracker/video_pose_data.py:75-151.  The mp4 decode itself needs
PyAV, which this image lacks; the same frames are read from an uncompressed .mono file."""
import importlib.util
import json

import numpy as np
import pytest

from umetrack_amd import eval as ev
from umetrack_amd.lib.common.hand import load_generic_hand_model
from umetrack_amd.lib.tracker import video_pose_data as vpd


def _write(tmp_path, n=5, **kw):
    hm = load_generic_hand_model()
    rec = ev.SyntheticRecording(n, 3, hm, **kw)
    p = str(tmp_path / "rec0.mono")
    vpd.write_recording(p, rec._frames, rec.cameras, rec.camera_angles, hm, rec.joint_angles, rec.wrist_transforms,
                        rec.hand_confidences)
    return rec, p


def test_stream_round_trip(tmp_path):
    rec, p = _write(tmp_path, drop_left_every=2)
    stream = vpd.SyncedImagePoseStream(p)
    assert len(stream) == len(rec) == 5 and stream.camera_angles == [0.0] * 4
    for (fa, ga), (fb, gb) in zip(rec, stream):
        assert sorted(ga) == sorted(gb)
        for h in ga:
            np.testing.assert_array_equal(ga[h].joint_angles, gb[h].joint_angles.astype(np.float32))
            np.testing.assert_array_equal(ga[h].wrist_xform, gb[h].wrist_xform.astype(np.float32))
            assert gb[h].hand_confidence > 0
        assert len(fb.views) == 4
        for va, vb in zip(fa.views, fb.views):
            np.testing.assert_array_equal(va.image, vb.image)
            assert vb.image.shape == (480, 640) and type(vb.camera) is type(va.camera)
            np.testing.assert_allclose(vb.camera.camera_to_world_xf, va.camera.camera_to_world_xf)
            np.testing.assert_allclose(vb.camera.f, va.camera.f)
            np.testing.assert_allclose(list(vb.camera.distort), list(va.camera.distort))
    hm = stream.hand_model
    assert tuple(hm.joint_rest_positions.shape) == (22, 3) and float(hm.hand_scale) == 1.0


def test_untracked_cameras_and_bad_files(tmp_path):
    rec, p = _write(tmp_path)
    js = json.load(open(p[:-5] + ".json"))
    js["camera_to_world_transforms"][2] = np.zeros((4, 4, 4)).tolist()      # rig pose lost in frame 2 ...
    json.dump(js, open(p[:-5] + ".json", "w"))
    with pytest.raises(AssertionError, match="Cameras are not tracked"):     # ... but hands are labelled
        list(vpd.SyncedImagePoseStream(p))
    js["hand_confidences"][2] = [0.0, 0.0]
    json.dump(js, open(p[:-5] + ".json", "w"))
    frames = list(vpd.SyncedImagePoseStream(p))
    assert frames[2][1] == {} and len(frames[1][1]) == 2
    with open(p, "ab") as fh:
        fh.write(b"\0" * 17)
    with pytest.raises(ValueError, match="whole number"):
        vpd.SyncedImagePoseStream(p)


@pytest.mark.skipif(importlib.util.find_spec("av") is not None, reason="PyAV present")
def test_mp4_needs_pyav(tmp_path):
    rec, p = _write(tmp_path)
    mp4 = p[:-5] + ".mp4"
    open(mp4, "wb").close()
    with pytest.raises(ImportError, match="PyAV"):
        vpd.SyncedImagePoseStream(mp4)
