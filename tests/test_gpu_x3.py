"""UMT_PREC_FP32_BF16X3 ("fp32x3", conv_x3.hip) through the C ABI.

Every tensor stays fp32; inside the stride-1 3x3 backbone layers an fp32 operand is the exact sum of three
bf16 pieces and a product is six bf16 matrix instructions accumulated in fp32 (dropped cross terms
<= 2^-23 of the product).  The mode is not bit-identical to "fp32" (an fmaf chain), so it is held to the SAME
tolerances as the fp32 parity mode against the reference's goldens (test_gpu_parity.py):
  wrist translation 1e-5 m (0.01 mm), joint angles 2e-5 rad, feature maps 2e-4 of the max magnitude,
and its distance from the float64 oracle is compared with the fp32 kernels' own distance.
"""
import numpy as np
import pytest
import torch

from oracle import model as om
from umetrack_amd import arch
from test_gpu_parity import _frame, _rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models(golden_sequence):
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    _, seed = golden_sequence
    state = arch.make_synthetic_state(seed)
    return model_from_state_dict(state).to("cuda"), state


@pytest.mark.parametrize("n", [1, 5, 16, 67, 460])
def test_x3_backbone_is_as_close_to_float64_as_the_fp32_kernels(models, n):
    """A layer runs on the split-operand kernels from 16 384 pixels per launch on (below that the fp32 kernels' few-pixel
    forms are faster: the tracker's per-frame call): none at 1 and 5 crops, stage 1 at 16, stages 1-2 at 67, all at 460."""
    gm, state = models
    img = torch.rand(n, 96, 96, generator=torch.Generator().manual_seed(200 + n))
    f32 = gm.backbone(img.cuda()).cpu().numpy()
    gm.precision = "fp32x3"
    try:
        x3 = gm.backbone(img.cuda()).cpu().numpy()
        x3b = gm.backbone(img.cuda()).cpu().numpy()
    finally:
        gm.precision = "fp32"
    assert np.array_equal(x3, x3b)                       # deterministic
    assert np.array_equal(x3, f32) == (n * 48 * 48 < 16384)      # the mode really runs the split kernels
    sd = om.to_torch_state(state, torch.float64)
    with torch.no_grad():
        truth = om.backbone(sd, img.double()).numpy()
    e32, e3 = _rel(f32, truth), _rel(x3, truth)
    rms32 = float(np.sqrt(np.mean((f32 - truth) ** 2)))
    rms3 = float(np.sqrt(np.mean((x3 - truth) ** 2)))
    print(f"n={n}: max rel err vs float64: fp32 {e32:.2e}, fp32x3 {e3:.2e}; rms {rms32:.2e} / {rms3:.2e}")
    assert e3 < 2e-4                                     # the fp32 mode's feature tolerance
    assert e3 < 3 * e32 + 1e-6 and rms3 < 3 * rms32 + 1e-9


def test_x3_sequence_matches_reference_goldens(models, golden_sequence):
    """The reference's own outputs for the 6-step sequence, at the fp32 mode's tolerances."""
    gm, _ = models
    steps, _ = golden_sequence
    gm.reset_memory()
    gm.precision = "fp32x3"
    worst_t = worst_a = 0.0
    try:
        for st in steps:
            name = str(st["name"])
            fd, desc, skel = _frame(st)
            out = gm.regress_pose_use_skeleton(fd, desc, skel) if skel is not None \
                else gm.regress_pose_pred_skel_scale(fd, desc)
            ja, wx = out.joint_angles.cpu().numpy(), out.wrist_xfs.cpu().numpy()
            worst_a = max(worst_a, float(np.abs(ja - st["joint_angles"]).max()))
            worst_t = max(worst_t, float(np.abs(wx[:, :3, 3] - st["wrist_xfs"][:, :3, 3]).max()))
            assert np.abs(ja - st["joint_angles"]).max() < 2e-5, name
            assert np.abs(wx[:, :3, 3] - st["wrist_xfs"][:, :3, 3]).max() < 1e-5, name          # metres
            assert np.abs(wx[:, :3, :3] - st["wrist_xfs"][:, :3, :3]).max() < 2e-5, name
            mem, _ = gm.memory_state()
            assert _rel(mem.cpu().numpy(), st["memory"]) < 2e-4, name
    finally:
        gm.precision = "fp32"
        gm.reset_memory()
    print(f"fp32x3 vs reference goldens: wrist {worst_t * 1e3:.5f} mm, joint angles {worst_a:.2e} rad")


def test_x3_random_batch_within_a_hundredth_of_a_millimetre_of_the_oracle():
    """64 hands, 3 steps with memory, landmarks after forward kinematics: fp32x3 vs the fp32 CPU oracle."""
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.common.hand_skinning import skin_landmarks
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    from scipy.spatial.transform import Rotation
    state = arch.make_synthetic_state(78)
    gm = model_from_state_dict(state).to("cuda")
    gm.precision = "fp32x3"
    o32 = om.OracleModel(state, torch.float32)
    rng = np.random.default_rng(6)
    bs = 64
    hm = load_generic_hand_model()
    skel = InputSkeletonData(hm.joint_rotation_axes, hm.joint_rest_positions * 0.001)
    for step in range(3):
        img = torch.from_numpy(rng.integers(0, 256, size=(2 * bs, 96, 96)).astype(np.float32) / np.float32(255))
        K = torch.zeros(2 * bs, 3, 3)
        K[:, 0, 0] = K[:, 1, 1] = torch.from_numpy(rng.uniform(150, 400, 2 * bs).astype(np.float32))
        K[:, 0, 2] = K[:, 1, 2] = 47.5
        K[:, 2, 2] = 1
        E = np.tile(np.eye(4, dtype=np.float32), (2 * bs, 1, 1))
        E[:, :3, :3] = Rotation.from_rotvec(rng.normal(size=(2 * bs, 3)) * 0.5).as_matrix()
        E[:, :3, 3] = rng.normal(size=(2 * bs, 3)) * 0.2
        fd = InputFrameData(img, K, torch.from_numpy(E))
        hand = torch.arange(bs) % 2
        desc = InputFrameDesc(torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(bs),
                              torch.full((bs,), step > 0), hand)
        got = gm.regress_pose_use_skeleton(fd, desc, skel)
        with torch.no_grad():
            want = o32.regress_pose_use_skeleton(fd, desc, skel)
        e_t = np.abs(got.wrist_xfs.cpu().numpy()[:, :3, 3] - want["wrist_xfs"].numpy()[:, :3, 3]).max()
        e_a = np.abs(got.joint_angles.cpu().numpy() - want["joint_angles"].numpy()).max()
        lm_got = skin_landmarks(hm, got.joint_angles, got.wrist_xfs, hand_idx=hand.cuda(), mirror_right=True,
                                wrist_translation_scale=1000.0).cpu().numpy()
        lm_want = skin_landmarks(hm, want["joint_angles"].cuda(), want["wrist_xfs"].cuda(), hand_idx=hand.cuda(),
                                 mirror_right=True, wrist_translation_scale=1000.0).cpu().numpy()
        e_lm = np.abs(lm_got - lm_want).max()          # millimetres
        print(f"step {step}: wrist {e_t * 1e3:.5f} mm, angles {e_a:.2e} rad, landmarks {e_lm:.5f} mm")
        assert e_t < 1e-5 and e_a < 2e-5 and e_lm < 0.01


@pytest.mark.parametrize("head_name", ["known", "scale"])
def test_x3_head_layers_on_large_batches_stay_at_fp32_rounding(monkeypatch, head_name):
    """From 8192 head pixels on (228 hands) the bf16 / fp32x3 modes of umt_model_forward run the known-skeleton regressor's
    3x3 layers on the split-operand kernels too (UMT_HEAD_NOX3 keeps them on the fp32 kernels; the scale head's 72-channel
    rows are not a multiple of the kernels' 16-channel groups and always stay on the fp32 kernels).  256 hands: same
    backbone bits, so every difference is the head's - it must be there for the known-skeleton head (the switch works),
    absent for the scale head, and both stay within the fp32 parity mode's own tolerances against the CPU oracle."""
    from umetrack_amd.lib.common.hand import load_generic_hand_model
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    from scipy.spatial.transform import Rotation
    state = arch.make_synthetic_state(41)
    monkeypatch.delenv("UMT_HEAD_NOX3", raising=False)
    split = model_from_state_dict(state).to("cuda")
    monkeypatch.setenv("UMT_HEAD_NOX3", "1")
    plain = model_from_state_dict(state).to("cuda")        # the switch is read per model handle
    split.precision = plain.precision = "fp32x3"
    rng = np.random.default_rng(8)
    bs = 256
    hm = load_generic_hand_model()
    skel = InputSkeletonData(hm.joint_rotation_axes, hm.joint_rest_positions * 0.001)
    img = torch.from_numpy(rng.integers(0, 256, size=(2 * bs, 96, 96)).astype(np.float32) / np.float32(255))
    K = torch.zeros(2 * bs, 3, 3)
    K[:, 0, 0] = K[:, 1, 1] = torch.from_numpy(rng.uniform(150, 400, 2 * bs).astype(np.float32))
    K[:, 0, 2] = K[:, 1, 2] = 47.5
    K[:, 2, 2] = 1
    E = np.tile(np.eye(4, dtype=np.float32), (2 * bs, 1, 1))
    E[:, :3, :3] = Rotation.from_rotvec(rng.normal(size=(2 * bs, 3)) * 0.5).as_matrix()
    E[:, :3, 3] = rng.normal(size=(2 * bs, 3)) * 0.2
    fd = InputFrameData(img, K, torch.from_numpy(E))
    desc = InputFrameDesc(torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(bs),
                          torch.zeros(bs, dtype=torch.bool), torch.arange(bs) % 2)
    o32 = om.OracleModel(state, torch.float32)
    if head_name == "known":
        a, b = split.regress_pose_use_skeleton(fd, desc, skel), plain.regress_pose_use_skeleton(fd, desc, skel)
        with torch.no_grad():
            want = o32.regress_pose_use_skeleton(fd, desc, skel)
    else:
        a, b = split.regress_pose_pred_skel_scale(fd, desc), plain.regress_pose_pred_skel_scale(fd, desc)
        with torch.no_grad():
            want = o32.regress_pose_pred_skel_scale(fd, desc)
    assert torch.equal(a.joint_angles, b.joint_angles) == (head_name == "scale"), "which kernels ran the head's 3x3 layers"
    for got in (a, b):
        e_t = np.abs(got.wrist_xfs.cpu().numpy()[:, :3, 3] - want["wrist_xfs"].numpy()[:, :3, 3]).max()
        e_a = np.abs(got.joint_angles.cpu().numpy() - want["joint_angles"].numpy()).max()
        assert e_t < 1e-5 and e_a < 2e-5, (e_t, e_a)
    split.check_status()
