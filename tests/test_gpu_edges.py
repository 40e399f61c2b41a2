"""Edge cases through the C ABI: tile-boundary batch sizes, single crops, error returns, determinism."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import model as om
from umetrack_amd import _native as nat
from umetrack_amd import arch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    state = arch.make_synthetic_state(3)
    return model_from_state_dict(state).to("cuda"), om.to_torch_state(state)


@pytest.mark.parametrize("n", [1, 3, 7, 57])
def test_backbone_odd_batch_sizes(pair, n):
    """Pixel counts that do not fill the 128-pixel GEMM tiles (36*n, 144*n ... rows)."""
    gm, sd = pair
    rng = np.random.default_rng(n)
    img = torch.from_numpy(rng.uniform(0, 1, size=(n, 96, 96)).astype(np.float32))
    got = gm.backbone(img).cpu()
    with torch.no_grad():
        ref = om.backbone(sd, img)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-4
    # the same crop gives the same features whatever its position in the batch
    again = gm.backbone(torch.cat([img[-1:], img])).cpu()
    assert torch.equal(again[0], got[-1])


def test_forward_is_deterministic_and_batch_invariant(pair):
    from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model
    from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    gm, _ = pair
    rng = np.random.default_rng(0)
    bs = 9
    img = torch.from_numpy(rng.uniform(0, 1, size=(2 * bs, 96, 96)).astype(np.float32))
    K = torch.tensor([[300.0, 0, 47.5], [0, 300.0, 47.5], [0, 0, 1]]).repeat(2 * bs, 1, 1)
    E = torch.eye(4).repeat(2 * bs, 1, 1)
    E[:, :3, 3] = torch.from_numpy(rng.normal(size=(2 * bs, 3)).astype(np.float32)) * 0.1
    hm = scaled_hand_model(load_generic_hand_model(), 0.001)
    skel = InputSkeletonData(hm.joint_rotation_axes, hm.joint_rest_positions)

    def run(sel):
        c = torch.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
        k = len(sel)
        gm.reset_memory()
        return gm.regress_pose_use_skeleton(
            InputFrameData(img[c], K[c], E[c]),
            InputFrameDesc(torch.arange(k)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(k),
                           torch.zeros(k, dtype=torch.bool), torch.zeros(k, dtype=torch.long)), skel)
    full = run(torch.arange(bs))
    full2 = run(torch.arange(bs))
    assert torch.equal(full.wrist_xfs, full2.wrist_xfs) and torch.equal(full.joint_angles, full2.joint_angles)
    part = run(torch.tensor([4, 2]))
    assert torch.equal(part.wrist_xfs.cpu(), full.wrist_xfs.cpu()[[4, 2]])


def test_error_codes_instead_of_faults(pair):
    gm, _ = pair
    L = nat.lib()
    img = torch.zeros(2, 96, 96, device="cuda")
    out = torch.zeros(2, 72, 6, 6, device="cuda")
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    st = nat.stream_ptr(gm.device)
    rc = L.umt_backbone_forward(gm._h, nat.ptr(img), 2, nat.PREC_FP32, nat.ptr(out), nat.ptr(small), small.numel(), st)
    assert rc == 5 and b"workspace" in L.umt_last_error()
    ws = torch.empty(L.umt_backbone_workspace_bytes(2), dtype=torch.uint8, device="cuda")
    rc = L.umt_backbone_forward(gm._h, nat.ptr(img), 2, nat.PREC_BF16 + 7, nat.ptr(out), nat.ptr(ws), ws.numel(), st)
    assert rc != 0
    rc = L.umt_backbone_forward(gm._h, nat.ptr(img), 0, nat.PREC_FP32, nat.ptr(out), nat.ptr(ws), ws.numel(), st)
    assert rc == 1
    with pytest.raises(ValueError, match="96,96"):
        gm.backbone(torch.zeros(2, 64, 64))
    # a model that was never finalized refuses to run
    h = ctypes.c_void_p()
    nat.check(L.umt_model_create(ctypes.byref(h)))
    rc = L.umt_backbone_forward(h, nat.ptr(img), 2, 0, nat.ptr(out), nat.ptr(ws), ws.numel(), st)
    assert rc == 2 and b"not finalized" in L.umt_last_error()
    L.umt_model_destroy(h)
    # malformed sample ranges poison their rows with NaN instead of reading out of bounds
    feats = torch.zeros(4, 72, 6, 6)
    K = torch.eye(3).repeat(4, 1, 1) * 200
    E = torch.eye(4).repeat(4, 1, 1)
    fused = gm.fuse_views(feats, K, E, torch.tensor([[0, 2], [2, 9]]))
    assert torch.isfinite(fused[0]).all() and torch.isnan(fused[1]).all()
    # empty batches are no-ops for the pointwise entry points
    assert L.umt_fk_landmarks(None, None, None, None, None, None, None, 1, None, 0, ctypes.c_float(1.0), 0, None, None, st) == 0
    assert L.umt_warp_crops(None, None, None, None, None, 0, 96, 96, 0, None, st) == 0


def test_memory_reserve_grows_and_keeps_rows(pair):
    gm, _ = pair
    gm.reset_memory()
    f = torch.randn(2, 72, 6, 6)
    E = torch.eye(4).repeat(2, 1, 1)
    gm.temporal_step(f, E, torch.tensor([0, 1]), torch.tensor([False, False]))
    mem1, _ = gm.memory_state()
    gm.temporal_step(f[:1], E[:1], torch.tensor([40]), torch.tensor([False]))       # grows to 41 rows
    mem2, prev2 = gm.memory_state()
    assert mem2.shape[0] == 41 and torch.equal(mem2[:2], mem1) and float(mem2[2:40].abs().max()) == 0
    assert torch.equal(prev2[40].cpu(), torch.eye(4))


@pytest.mark.gpu
def test_layer_timing_covers_every_backbone_launch():
    """umt_model_layer_timing_*: 26 launches per backbone pass (22 in bf16 mode), their FLOP add up to SURVEY 8(d)'s 969.23 MFLOP per
    crop, their times to the backbone time of umt_model_timing_read."""
    import torch
    from umetrack_amd import arch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    dev = torch.device("cuda", 0)
    m = model_from_state_dict(arch.make_synthetic_state(0)).to(dev)
    n = 256
    img = torch.rand(n, 96, 96, device=dev)
    for prec in ("fp32", "bf16", "fp32x3"):
        m.precision = prec
        m.backbone(img)                     # warm-up
        m.layer_timing_enable(True)
        for _ in range(2):
            m.backbone(img)
        rows = m.layer_timing_read()
        m.layer_timing_enable(False)
        # 26 launches; in bf16 mode both BasicBlocks of stage 1, the 32 -> 64 stride-2 layer and its down-sample branch
        # are ONE launch (stage1_depth_bf16)
        assert len(rows) == (22 if prec == "bf16" else 26), rows
        assert rows[0][0].startswith("stem") and rows[-1][0].startswith("proj")
        assert all(r[3] == 2 and r[2] > 0 for r in rows)
        flop = sum(r[1] for r in rows) / 2 / n
        assert abs(flop - 969.23e6) < 0.02e6, flop
    m.layer_timing_enable(True)
    assert m.layer_timing_read() == []
    m.layer_timing_enable(False)


@pytest.mark.gpu
def test_fp32_persistent_stage1_kernel_bit_identical_to_halo_kernel(tmp_path):
    """fp32 mode, 32 -> 32 layers of stage 1: from 57 crops up they run on a persistent LDS-weights kernel of
    conv_thin.hip (conv3x3_pipe32_f32; conv3x3_wlds32_f32 with UMT_CONV_NOPIPE32=1); with UMT_CONV_NOTHIN_F32=1 (read once per process, hence the child process) on
    conv3x3s1_halo_f32.  Same k order and epilogue arithmetic, so the backbone features must agree bit for bit; 131
    crops leave the last 128-pixel tiles of the walk ragged, 57 is the switch-over size."""
    import os
    import subprocess
    import sys
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "halo.npy"
    body = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from umetrack_amd import arch\n"
        "from umetrack_amd.lib.models.model_loader import model_from_state_dict\n"
        "gm = model_from_state_dict(arch.make_synthetic_state(5)).to('cuda')\n"
        "img = torch.rand(131, 96, 96, generator=torch.Generator().manual_seed(77))\n"
        "res = np.concatenate([gm.backbone(img[:57].cuda()).cpu().numpy(), gm.backbone(img.cuda()).cpu().numpy()])\n")
    env = dict(os.environ, UMT_CONV_NOTHIN_F32="1")
    subprocess.run([sys.executable, "-c", body + f"np.save({str(out)!r}, res)\n"], check=True, env=env, timeout=110)
    gm = model_from_state_dict(arch.make_synthetic_state(5)).to("cuda")
    img = torch.rand(131, 96, 96, generator=torch.Generator().manual_seed(77))
    got = np.concatenate([gm.backbone(img[:57].cuda()).cpu().numpy(), gm.backbone(img.cuda()).cpu().numpy()])
    assert np.array_equal(got, np.load(out))


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [1, 37, 256])
def test_fused_1x1_chains_bit_identical_to_three_launches(bs, monkeypatch):
    """View fusion (144 -> 108 -> 72 -> 72) and the temporal module (90 -> 90 -> 90 -> 90) run as one kernel each
    (csrc/conv_chain.hip: operands swapped, a layer's accumulator registers are the next layer's operand).  With
    UMT_CONV_NOCHAIN=1 (read per model handle at umt_model_create) they run as three conv_igemm_f32 launches.  Same k
    order and epilogue arithmetic: fused features, temporal output and the memory state left behind must agree bit for
    bit; bs = 1 and 37 leave the last 32-pixel wave tile ragged (36 and 1332 rows) and take the few-rows form
    (conv1x1_chain3_small_f32: a workgroup per 32-row tile, a wave per cout tile, layers exchanged through LDS), bs = 256
    (288 tiles) the persistent one."""
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    g = torch.Generator().manual_seed(100 + bs)
    n = 2 * bs
    feats = torch.randn(n, 72, 6, 6, generator=g).cuda()
    K = torch.tensor([[120.0, 0, 48], [0, 120.0, 48], [0, 0, 1]]).repeat(n, 1, 1)
    K[:, 0, 0] += torch.rand(n, generator=g) * 20
    E = torch.eye(4).repeat(n, 1, 1)
    E[:, :3, 3] = torch.randn(n, 3, generator=g) * 0.05
    sr = torch.stack([torch.arange(0, n, 2), torch.arange(2, n + 2, 2)], 1)
    midx = torch.arange(bs)

    def run():
        gm = model_from_state_dict(arch.make_synthetic_state(9)).to("cuda")
        fused = gm.fuse_views(feats, K, E, sr)
        t0 = gm.temporal_step(fused, E[0::2].contiguous(), midx, torch.zeros(bs, dtype=torch.bool), required_rows=bs)
        t1 = gm.temporal_step(fused * 0.5, E[1::2].contiguous(), midx, torch.ones(bs, dtype=torch.bool), required_rows=bs)
        gm.check_status()
        return [x.cpu() for x in (fused, t0, t1)]

    got = run()
    monkeypatch.setenv("UMT_CONV_NOCHAIN", "1")
    ref = run()
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.gpu
def test_persistent_stride2_kernel_bit_identical_to_one_tile_kernel(monkeypatch):
    """fp32 mode: the three fused stride-2 + down-sample layers run on the persistent kernel conv3x3s2p_ds_f32 (a workgroup
    walks its XCD's tiles and fetches the next tile's input under the epilogue); with UMT_CONV_NOS2P=1 (read per model handle)
    on conv3x3s2_ds_f32, one tile per workgroup.  Same tiles, k order and epilogue: features must agree bit for bit.
    3 crops leave most workgroups without a tile on the last layer, 131 make the last tiles ragged."""
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    img = torch.rand(131, 96, 96, generator=torch.Generator().manual_seed(78)).cuda()

    def run():
        gm = model_from_state_dict(arch.make_synthetic_state(6)).to("cuda")
        return torch.cat([gm.backbone(img[:3]), gm.backbone(img)]).cpu()

    got = run()
    monkeypatch.setenv("UMT_CONV_NOS2P", "1")
    ref = run()
    assert torch.isfinite(got).all() and torch.equal(got, ref)


@pytest.mark.gpu
def test_fp32_single_role_stage1_kernel_bit_identical_to_two_group_kernel(monkeypatch):
    """fp32 mode, 32 -> 32 layers of stage 1: conv3x3_pipe32_f32 (every wave carries its own epilogue, halo stores and loads
    between its MFMAs; operands swapped, 256-pixel tiles) against conv3x3_wlds32_f32 (UMT_CONV_NOPIPE32=1, read per model
    handle; two groups of waves alternating MFMA and epilogue phases, 128-pixel tiles).  Same k order and epilogue
    arithmetic: bit-identical features.  57 crops is the switch-over size, 131 leave the last tiles of the walk ragged,
    900 give every workgroup several tiles (the steady state of the pipeline, both LDS buffers in use)."""
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    img = torch.rand(900, 96, 96, generator=torch.Generator().manual_seed(79)).cuda()

    def run():
        gm = model_from_state_dict(arch.make_synthetic_state(7)).to("cuda")
        return torch.cat([gm.backbone(img[:57]), gm.backbone(img[:131]), gm.backbone(img)]).cpu()

    got = run()
    monkeypatch.setenv("UMT_CONV_NOPIPE32", "1")
    ref = run()
    assert torch.isfinite(got).all() and torch.equal(got, ref)


@pytest.mark.gpu
def test_bf16_one_role_64_channel_kernel_bit_identical_to_two_group_kernel(monkeypatch):
    """bf16 mode, 64 -> 64 layers of stage 2: conv3x3_pipe64_bf16 (one role per wave, 2 x 2 MFMA tiles, operands swapped,
    256-pixel tiles, accumulators of consecutive tiles in alternating register sets, quads traded between the wave halves
    for 16-byte stores) against conv3x3_wlds64_bf16 (UMT_CONV_NOPIPE64=1, read per model handle).  Same k order and
    epilogue arithmetic: bit-identical features.  3 crops leave most workgroups without a tile, 131 leave the last tiles
    ragged, 900 reach the steady state of the walk (several tiles per workgroup, odd and even counts)."""
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    img = torch.rand(900, 96, 96, generator=torch.Generator().manual_seed(80)).cuda()

    def run():
        gm = model_from_state_dict(arch.make_synthetic_state(8)).to("cuda")
        gm.precision = "bf16"
        return torch.cat([gm.backbone(img[:3]), gm.backbone(img[:131]), gm.backbone(img)]).float().cpu()

    got = run()
    monkeypatch.setenv("UMT_CONV_NOPIPE64", "1")
    ref = run()
    assert torch.isfinite(got).all() and torch.equal(got, ref)


@pytest.mark.gpu
def test_fp32_few_pixel_kernel_bit_identical_to_halo_kernel(tmp_path):
    """fp32 mode with few crops (the tracker's per-frame call is 4): the 64/128/256-channel stride-1 layers run on
    conv_small.hip (16x16x4 matrix instructions, k-slots in the channel order of four 32x32x2 instructions); with
    UMT_CONV_NOSMALL=1 (read once per process, hence the child process) on conv3x3s1_halo_f32<4,1,1,1>.  Same chain per
    output, so the backbone features must agree bit for bit: 1, 4 and 13 crops (ragged 16-pixel tiles), and 40 crops,
    where the 64-channel layers are past the switch-over and the 256-channel layers are not.  The three fused stride-2
    launches take conv3x3s2_small_f32 up to 16 384 pixel x cout-tile units (1, 4 and 13 crops) and conv3x3s2p_ds_f32 beyond
    (40 crops); the same switch selects between them.  The regressor's four 3x3
    layers (80 -> 76 channels padded to 96: the last workgroup along cout has two waves without channels) take the same
    kernel for few hands: its 62 raw outputs for 1, 2 and 5 hands, both heads, must agree bit for bit too."""
    import os
    import subprocess
    import sys
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "halo.npy"
    sizes = (1, 4, 13, 40)
    body = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from umetrack_amd import arch\n"
        "from umetrack_amd.lib.models.model_loader import model_from_state_dict\n"
        "gm = model_from_state_dict(arch.make_synthetic_state(3)).to('cuda')\n"
        "img = torch.rand(40, 96, 96, generator=torch.Generator().manual_seed(77))\n"
        "from types import SimpleNamespace as NS\n"
        "g = torch.Generator().manual_seed(79)\n"
        "tmp = torch.randn(5, 72, 6, 6, generator=g); E0 = torch.eye(4).repeat(5, 1, 1); hidx = torch.tensor([0, 1, 1, 0, 1])\n"
        "skel = NS(joint_rotation_axes=torch.randn(5, 22, 3, generator=g), joint_rest_positions=torch.randn(5, 22, 3, generator=g) * 0.05)\n"
        "raws = []\n"
        "for k in (1, 2, 5):\n"
        "    sk = NS(joint_rotation_axes=skel.joint_rotation_axes[:k], joint_rest_positions=skel.joint_rest_positions[:k])\n"
        "    raws.append(gm.regress_decode(0, tmp[:k], E0[:k], hidx[:k], sk)[1].cpu().numpy().ravel())\n"
        "    raws.append(gm.regress_decode(1, tmp[:k], E0[:k], hidx[:k])[1].cpu().numpy().ravel())\n"
        f"np.save({str(out)!r}, np.concatenate([gm.backbone(img[:k].cuda()).cpu().numpy().ravel() for k in {sizes!r}] + raws))\n")
    subprocess.run([sys.executable, "-c", body], check=True, env=dict(os.environ, UMT_CONV_NOSMALL="1"), timeout=110)
    from umetrack_amd import arch
    gm = model_from_state_dict(arch.make_synthetic_state(3)).to("cuda")
    img = torch.rand(40, 96, 96, generator=torch.Generator().manual_seed(77))
    from types import SimpleNamespace as NS
    g = torch.Generator().manual_seed(79)
    tmp = torch.randn(5, 72, 6, 6, generator=g)
    E0 = torch.eye(4).repeat(5, 1, 1)
    hidx = torch.tensor([0, 1, 1, 0, 1])
    skel = NS(joint_rotation_axes=torch.randn(5, 22, 3, generator=g), joint_rest_positions=torch.randn(5, 22, 3, generator=g) * 0.05)
    raws = []
    for k in (1, 2, 5):
        sk = NS(joint_rotation_axes=skel.joint_rotation_axes[:k], joint_rest_positions=skel.joint_rest_positions[:k])
        raws.append(gm.regress_decode(0, tmp[:k], E0[:k], hidx[:k], sk)[1].cpu().numpy().ravel())
        raws.append(gm.regress_decode(1, tmp[:k], E0[:k], hidx[:k])[1].cpu().numpy().ravel())
    got = np.concatenate([gm.backbone(img[:k].cuda()).cpu().numpy().ravel() for k in sizes] + raws)
    want = np.load(out)
    assert np.isfinite(got).all() and np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.gpu
def test_matrix_pipe_stem_bit_identical_to_vector_stem(tmp_path, monkeypatch):
    """The stem (conv 1 -> 32 + BN + ReLU + max-pool) runs on v_mfma_f32_32x32x2_f32 (stem_mfma_kernel: K = 9 padded to 10,
    max-pool by DPP inside the accumulator quads); with UMT_STEM_VALU=1 (read once per process, hence the child process)
    on the vector pipe.  The matrix instruction is the same k-ordered fmaf chain, so fp32 and bf16 backbone features must
    agree bit for bit, and a NaN / inf pixel must poison exactly the same pooled pixels.  (The bf16 mode's default stem
    multiplies split operands on the bf16 matrix instructions instead - next test; UMT_STEM_NOSPLIT keeps the chain.)"""
    import os
    import subprocess
    import sys
    import torch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "valu.npy"
    body = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from umetrack_amd import arch\n"
        "from umetrack_amd.lib.models.model_loader import model_from_state_dict\n"
        "gm = model_from_state_dict(arch.make_synthetic_state(4)).to('cuda')\n"
        "img = torch.rand(1030, 96, 96, generator=torch.Generator().manual_seed(78))\n"
        "img[3, 0, 0] = float('nan'); img[5, 95, 95] = float('inf'); img[6, 40, 41] = float('nan'); img[1027, 57, 8] = float('nan')\n"
        "res = []\n"
        "for prec in ('fp32', 'bf16'):\n"
        "    gm.precision = prec\n"
        "    res += [gm.backbone(img[:k].cuda()).cpu().numpy() for k in (2, 70, 1030)]\n"
        f"np.save({str(out)!r}, np.concatenate(res))\n")
    subprocess.run([sys.executable, "-c", body], check=True, env=dict(os.environ, UMT_STEM_VALU="1"), timeout=110)
    from umetrack_amd import arch
    monkeypatch.setenv("UMT_STEM_NOSPLIT", "1")            # read per model handle
    gm = model_from_state_dict(arch.make_synthetic_state(4)).to("cuda")
    img = torch.rand(1030, 96, 96, generator=torch.Generator().manual_seed(78))
    img[3, 0, 0] = float("nan")
    img[5, 95, 95] = float("inf")
    img[6, 40, 41] = float("nan")
    img[1027, 57, 8] = float("nan")        # 1024 crops and more: a workgroup walks a crop's 12 bands (double-buffered rows)
    res = []
    for prec in ("fp32", "bf16"):
        gm.precision = prec
        res += [gm.backbone(img[:k].cuda()).cpu().numpy() for k in (2, 70, 1030)]
    got, want = np.concatenate(res), np.load(out)
    assert np.isnan(got).any() and np.isfinite(got[:2]).all()
    assert np.array_equal(got, want, equal_nan=True), float(np.nanmax(np.abs(got - want)))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [3, 1030])
def test_bf16_stem_on_split_operands_equals_the_rounded_fp32_stem(monkeypatch, n):
    """The bf16 mode's stem (umt_stem_forward, UMT_PREC_BF16) multiplies (hi, lo) bf16 pairs of pixels and weights on
    v_mfma_f32_32x32x16_bf16 and drops the lo x lo term: 2^-16 of a product, against the 2^-9 rounding of the stored bf16
    output.  So it must equal the bf16 rounding of the fp32 stem except where the fp32 value sits within that distance of a
    rounding boundary (0.13 % of the outputs) and stay within half a bf16 step of the fp32 value everywhere; NaN / inf
    pixels poison the same pooled pixels; and UMT_STEM_NOSPLIT gives the rounded fp32 stem exactly."""
    import torch
    from umetrack_amd import arch
    from umetrack_amd.lib.models.model_loader import model_from_state_dict
    monkeypatch.delenv("UMT_STEM_NOSPLIT", raising=False)
    gm = model_from_state_dict(arch.make_synthetic_state(4)).to("cuda")
    monkeypatch.setenv("UMT_STEM_NOSPLIT", "1")
    chain = model_from_state_dict(arch.make_synthetic_state(4)).to("cuda")
    img = torch.rand(n, 96, 96, generator=torch.Generator().manual_seed(80))
    img[1, 0, 0] = float("nan")
    img[2, 95, 95] = float("inf")
    img[n - 1, 40, 41] = float("nan")
    ref = gm.stem(img)                                    # fp32 mode
    assert ref.dtype == torch.float32 and tuple(ref.shape) == (n, 48, 48, 32)
    gm.precision = chain.precision = "bf16"
    got, rounded = gm.stem(img), chain.stem(img)
    assert got.dtype == torch.bfloat16
    want = ref.to(torch.bfloat16)
    assert torch.equal(rounded.view(torch.int16), want.view(torch.int16))
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.isnan(got).any()
    ok = ~torch.isnan(want)
    g, w, r = got[ok].float(), want[ok].float(), ref[ok]
    frac = (g != w).float().mean().item()
    assert frac < 0.005, frac                             # measured 0.0013
    # within half a bf16 step of the fp32 value, plus the dropped term and the fp32 summation order: 2^-16 of the largest
    # output (measured 2^-17.5; relative to the value itself it is unbounded where the nine products cancel)
    assert ((g - r).abs() <= r.abs() * 2.0 ** -8 + r.abs().max() * 2.0 ** -16).all()
