"""BASELINE.json configs[2] at its full size (4096 frames x 2 hands x 2 views = 8192 hand-frames, 16 384 crops)
through the same driver as bench.py.  The CPU oracle cannot run this size in seconds, so the checks are
size-independent properties, all bit-exact unless stated:
  * determinism: two passes give identical landmarks;
  * a hand's result does not depend on the batch around it: the first 64 hands recomputed alone, and a whole
    4096-hand chunk recomputed in reversed order, reproduce the full-batch landmarks;
  * the first 32 hands agree with the CPU oracle to the 0.01 mm of the north star;
  * MPJPE / MPJPA from the GPU partial sums equal a float64 numpy evaluation of the downloaded landmarks
    (rtol 1e-12), and the partial vector is additive over a split of the recordings.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe():
    import bench
    p = bench.Pipeline(4096, 64, 2048, seed=0, device=torch.device("cuda", 0))
    p.step()
    torch.cuda.synchronize()
    return p


def _run_subset(p, sel):
    """Landmarks of the samples `sel` (indices into the workload) computed as a batch of their own."""
    from umetrack_amd.lib.common.hand_skinning import skin_landmarks
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc
    from umetrack_amd.lib.tracker import crop_cameras as gpu_crop
    from umetrack_amd.lib.tracker import warp as gpu_warp
    import umetrack_amd.synthetic as synthetic
    dev = p.dev
    sel = torch.as_tensor(sel, device=dev)
    k = sel.numel()
    crops_idx = torch.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
    pts = gpu_crop.crop_points_batched(p.hm, p.gt_angles[sel], p.gt_wrist[sel], p.hand_idx[sel], 63)
    cams = gpu_crop.gen_crop_cameras_batched(pts, p.rig_cams, p.rig_angles, p.hand_idx[sel], forced_views=p.views[sel])
    dst, K, E = cams.dst_cams.view(-1, 20), cams.intrinsics.view(-1, 3, 3), cams.extrinsics.view(-1, 4, 4)
    crops = gpu_warp.warp_crops(p.images, p.src_off[crops_idx], p.src_pitch[crops_idx], p.src_cams[crops_idx], dst)
    desc = InputFrameDesc((torch.arange(k, device=dev)[:, None] * 2 + torch.tensor([0, 2], device=dev)).contiguous(),
                          torch.arange(k, device=dev), torch.zeros(k, dtype=torch.bool, device=dev), p.hand_idx[sel])
    out = p.model.regress_pose_use_skeleton(InputFrameData(crops, K, E), desc, p.skel, required_rows=max(k, p.chunk))
    return skin_landmarks(p.hm, out.joint_angles, out.wrist_xfs, hand_idx=p.hand_idx[sel], mirror_right=True,
                          wrist_translation_scale=1000.0)


def test_full_batch_is_deterministic(pipe):
    first = pipe.pred_lm.clone()
    pipe.step()
    torch.cuda.synchronize()
    assert pipe.S == 8192 and pipe.pred_lm.numel() == 8192 * 63
    assert torch.equal(first, pipe.pred_lm)
    assert torch.isfinite(pipe.pred_lm).all()


def test_result_of_a_hand_does_not_depend_on_its_batch(pipe):
    full = pipe.pred_lm.view(-1, 21, 3)
    alone = _run_subset(pipe, np.arange(64))
    assert torch.equal(alone, full[:64])
    rev = _run_subset(pipe, np.arange(pipe.chunk)[::-1].copy())          # one whole 4096-hand chunk, reversed
    assert torch.equal(rev.flip(0), full[:pipe.chunk])


def test_first_hands_match_cpu_oracle(pipe):
    import bench
    _, cpu_pred, _ = bench.cpu_baseline(pipe, 32, budget_s=0.0)
    got = pipe.pred_lm.view(-1, 21, 3)[:32].cpu().numpy()
    assert np.abs(got - cpu_pred).max() < 0.01            # mm: the fp32 tolerance of BASELINE.json's north star


def test_metric_partials_match_numpy_and_are_additive(pipe):
    from umetrack_amd.lib.common import metric_utils as mu
    fin = mu.finalize_partials(pipe.partials)
    gt = pipe.gt_lm.cpu().numpy().astype(np.float64)      # (rec, 2, T, 21, 3)
    pr = pipe.pred_lm.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(gt - pr, axis=-1).mean(-1)
    np.testing.assert_allclose(fin.mpjpe, err.mean(), rtol=1e-12)
    acc = np.linalg.norm((pr[:, :, :-2] + pr[:, :, 2:] - 2 * pr[:, :, 1:-1]), axis=-1).mean(-1)
    np.testing.assert_allclose(fin.mpjpa, acc.mean(), rtol=1e-12)
    assert fin.n_valid == 8192 and fin.n_total == 8192
    h = pipe.gt_lm.shape[0] // 2
    a = mu.metric_partials(pipe.gt_lm[:h], pipe.pred_lm[:h], pipe.valid[:h])
    b = mu.metric_partials(pipe.gt_lm[h:], pipe.pred_lm[h:], pipe.valid[h:])
    np.testing.assert_allclose((a + b).cpu().numpy(), pipe.partials.cpu().numpy(), rtol=1e-12, atol=0)
