"""UMT_PREC_BF16 backbone (conv_bf16.hip) through the C ABI.

The reference has no reduced-precision mode, so nothing here is pinned to it ("parity unpinned"):
  * the bf16 kernels are checked against oracle.model.backbone_bf16, a CPU emulation that rounds
    the same tensors (weights, stored activations) to bfloat16 and accumulates wide.  The two
    differ only by fp32 accumulation order and the 1-ulp bf16 roundings that order can flip,
    tolerance 1.5 % of the feature map's max magnitude;
  * the cost of the mode is stated against the fp32 path.  It depends on the weights: over 8 weight
    seeds on the BASELINE configs[2] batch the per-coordinate landmark difference has mean
    0.19-0.68 mm, p99.9 0.81-4.43 mm and max 1.16-8.46 mm (profiles/r03_bf16_tolerance_8seeds.json,
    tools/bf16_tolerance.py).  The stated tolerance is 1.5 x the worst seed - mean 1.0 mm, p99.9 6.6 mm,
    max 12.7 mm (UmeTrackModel.BF16_LANDMARK_TOLERANCE_MM) - and is held here on two of those seeds
    (the mildest and the worst); on the golden sequence (one seed) the wrist stays within 1 mm.
    fp32 parity mode holds 0.01 mm; bf16 carries 8 significant bits through 26 layers.  Use fp32
    where the reference's numbers must be met.
"""
import numpy as np
import pytest
import torch

from oracle import model as om
from umetrack_amd import arch
from test_gpu_parity import _frame, _rel

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1.5e-2        # relative to max |feature|
WRIST_TOL_M = 1.0e-3
ANGLE_TOL_RAD = 2.5e-2


@pytest.fixture(scope="module")
def models(golden_sequence):
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    _, seed = golden_sequence
    state = arch.make_synthetic_state(seed)
    gm = model_from_state_dict(state).to("cuda")
    gm._umt_test_seed = seed
    return gm, state


def gm_seed(models):
    return models[0]._umt_test_seed


@pytest.mark.parametrize("n", [1, 5, 16])
def test_bf16_backbone_matches_emulation(models, n):
    gm, state = models
    g = torch.Generator().manual_seed(100 + n)
    img = torch.rand(n, 96, 96, generator=g)
    gm.precision = "bf16"
    try:
        got = gm.backbone(img.cuda()).cpu().numpy()
    finally:
        gm.precision = "fp32"
    sd = om.to_torch_state(state, torch.float64)
    with torch.no_grad():
        want = om.backbone_bf16(sd, img.double()).numpy()
        full = om.backbone(sd, img.double()).numpy()
    err = _rel(got, want)
    print(f"bf16 backbone n={n}: vs emulation {err:.2e}, emulation vs fp64 {_rel(want, full):.2e}")
    assert err < FEATURE_TOL
    # the mode really runs in reduced precision (guards against silently routing to the fp32 kernels)
    assert _rel(got, full) > 1e-4


def test_bf16_is_deterministic_and_leaves_fp32_untouched(models):
    gm, _ = models
    img = torch.rand(6, 96, 96, generator=torch.Generator().manual_seed(7)).cuda()
    ref32 = gm.backbone(img)
    gm.precision = "bf16"
    try:
        a, b = gm.backbone(img), gm.backbone(img)
    finally:
        gm.precision = "fp32"
    assert torch.equal(a, b)
    assert torch.equal(gm.backbone(img), ref32)
    with pytest.raises(ValueError):
        gm.precision = "fp8"


def test_bf16_end_to_end_cost_on_golden_sequence(models, golden_sequence):
    gm, _ = models
    steps, _ = golden_sequence
    worst_t = worst_a = 0.0
    for prec in ("fp32", "bf16"):
        gm.reset_memory()
        gm.precision = prec
        outs = []
        try:
            for st in steps:
                fd, desc, skel = _frame(st)
                out = gm.regress_pose_use_skeleton(fd, desc, skel) if skel is not None \
                    else gm.regress_pose_pred_skel_scale(fd, desc)
                outs.append((out.joint_angles.cpu().numpy(), out.wrist_xfs.cpu().numpy()))
        finally:
            gm.precision = "fp32"
        if prec == "fp32":
            base = outs
    for (ja32, wx32), (ja16, wx16) in zip(base, outs):
        worst_a = max(worst_a, float(np.abs(ja32 - ja16).max()))
        worst_t = max(worst_t, float(np.abs(wx32[:, :3, 3] - wx16[:, :3, 3]).max()))
    print(f"bf16 vs fp32 on the golden sequence: wrist {worst_t * 1e3:.3f} mm, joint angles {worst_a:.2e} rad")
    assert worst_t < WRIST_TOL_M
    assert worst_a < ANGLE_TOL_RAD
    gm.reset_memory()


@pytest.mark.parametrize("seed", [1, 3])
def test_bf16_landmark_difference_within_stated_tolerance(seed):
    """The stated tolerance (derived over 8 weight seeds at 8192 hand-frames) on a 1024-hand-frame batch of the same
    synthetic workload, for the mildest (1) and the worst (3) of those seeds."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from umetrack_amd.lib.models.umetrack_model import UmeTrackModel
    pipe = bench.Pipeline(512, 64, 512, seed=0, device=torch.device("cuda", 0), weight_seed=seed)
    pipe.model.precision = "fp32"
    pipe.step()
    p32 = pipe.pred_lm.clone()
    pipe.model.precision = "bf16"
    pipe.step()
    d = (pipe.pred_lm - p32).abs().reshape(-1).double()
    tol = UmeTrackModel.BF16_LANDMARK_TOLERANCE_MM
    got = {"mean": float(d.mean()), "p999": float(torch.quantile(d, 0.999)), "max": float(d.max())}
    print(f"seed {seed}: bf16 vs fp32 landmarks {got} mm (stated {tol})")
    assert torch.isfinite(pipe.pred_lm).all() and got["max"] > 0.0
    for k in tol:
        assert got[k] <= tol[k], (k, got, tol)


@pytest.mark.parametrize("switch", ["UMT_CONV_NOTHIN", "UMT_CONV_NODEPTH", "UMT_CONV_NOWALK"])
def test_bf16_persistent_kernels_bit_identical_to_per_layer_kernels(models, tmp_path, switch):
    """By default stage 1 and the 32 -> 64 stride-2 layer run depth-first in ONE launch (stage1_depth.hip: role waves and
    LDS rings), stages 2-3 on the persistent kernels of conv_thin.hip.  UMT_CONV_NODEPTH=1 (read once per process, hence
    the child process) runs stage 1 as two fused BasicBlock launches + the fused stride-2 launch instead;
    UMT_CONV_NOTHIN=1 runs every layer on conv3x3s1_halo_bf16 / conv3x3s2_ds_bf16.  Same arithmetic order everywhere, so
    the backbone features must agree bit for bit: for 1 and 7 crops (one crop per workgroup, pipeline fill and drain
    only), for 131 crops and for 300 (more crops than workgroups: crops streamed back to back through the rings).
    UMT_CONV_NOWALK=1 runs the 64 -> 128 stride-2 layer one tile per workgroup instead of on conv3x3s2_walk_bf16 (512
    workgroups walking the tiles with the next tile's input rows in flight); that kernel takes over above 512 tiles:
    675 at 300 crops (one or two tiles per workgroup), 2250 at 1000 crops (ragged walks of four and five tiles)."""
    import os
    import subprocess
    import sys
    gm, _ = models
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "ref.npy"
    sizes = (1, 7, 131, 300) + ((1000,) if switch == "UMT_CONV_NOWALK" else ())
    child = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from umetrack_amd import arch\n"
        "from umetrack_amd.lib.models.model_loader import model_from_state_dict\n"
        f"gm = model_from_state_dict(arch.make_synthetic_state({int(gm_seed(models))})).to('cuda')\n"
        "gm.precision = 'bf16'\n"
        f"img = torch.rand({max(sizes)}, 96, 96, generator=torch.Generator().manual_seed(31))\n"
        f"np.save({str(out)!r}, np.concatenate([gm.backbone(img[:k].cuda()).cpu().numpy() for k in {sizes!r}]))\n")
    env = dict(os.environ, **{switch: "1"})
    subprocess.run([sys.executable, "-c", child], check=True, env=env, timeout=110)
    img = torch.rand(max(sizes), 96, 96, generator=torch.Generator().manual_seed(31))
    gm.precision = "bf16"
    try:
        got = np.concatenate([gm.backbone(img[:k].cuda()).cpu().numpy() for k in sizes])
    finally:
        gm.precision = "fp32"
    want = np.load(out)
    assert np.isfinite(got).all()
    assert np.array_equal(got, want), float(np.abs(got - want).max())
