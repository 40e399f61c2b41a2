#!/usr/bin/env python3
"""Throughput of the UmeTrack hot path on MI355X: hand-frames/s on the synthetic 4-view x 2-hand
batch (BASELINE.json configs[2]).

One step = one pass over the whole batch: crop cameras from the hand poses (FK of 3 poses + look-at
geometry, on the device) -> warp 4-view fisheye frames into 2 crops per hand ->
network (fp32 MFMA, known-skeleton head) -> forward kinematics of prediction and ground truth ->
MPJPE/MPJPA/PCK partial sums -> one all-reduce of the 108-word partial vector (RCCL, N>1 only).
Source frames, cameras and poses are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded by the library
on the launch stream around the backbone convolutions; `cpu_baseline` times the CPU oracle
(oracle/, a port of the reference checked against its golden vectors) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from umetrack_amd import arch, dist as udist, synthetic  # noqa: E402
from umetrack_amd.lib.common import metric_utils as mu  # noqa: E402
from umetrack_amd.lib.common.hand import hand_model_to, load_generic_hand_model, scaled_hand_model  # noqa: E402
from umetrack_amd.lib.common.hand_skinning import skin_landmarks  # noqa: E402
from umetrack_amd.lib.models.model_loader import model_from_state_dict  # noqa: E402
from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData, UmeTrackModel  # noqa: E402
from umetrack_amd.lib.tracker import crop_cameras as gpu_crop  # noqa: E402
from umetrack_amd.lib.tracker import warp as gpu_warp  # noqa: E402

BACKBONE_FLOP_PER_CROP = 969.23e6      # 484.61 MMAC per 96x96 crop
MODEL_FLOP_PER_HAND = 1.958e9          # 2 views, known-skeleton head
FP32_MATRIX_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_PEAK_TFLOPS = 2516.6       # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E peak (about 6.3 TB/s is achievable)


def backbone_activation_bytes_per_crop(esz):
    """Algorithmic HBM bytes of the UNFUSED backbone per crop: input + output (+ residual) rows of each of its
    28 launches, activations of `esz` bytes (4 = fp32 rows, 2 = bf16 rows); weights (17 MB per call) left out."""
    rd, wr = 96 * 96 * 4, 48 * 48 * 32 * esz                      # stem: fp32 crop in, pooled rows out
    hw = 48
    for cin, cout, st, nb in [(32, 32, 1, 2), (32, 64, 2, 3), (64, 128, 2, 5), (128, 256, 2, 2)]:
        for blk in range(nb):
            ci, s = (cin, st) if blk == 0 else (cout, 1)
            ho = hw // s
            rd += hw * hw * ci * esz                               # conv1
            wr += ho * ho * cout * esz
            if blk == 0 and st != 1:                               # 1x1 stride-2 down-sample branch
                rd += ho * ho * ci * esz
                wr += ho * ho * cout * esz
            rd += 2 * ho * ho * cout * esz                         # conv2 reads conv1's output and the residual
            wr += ho * ho * cout * esz
            hw = ho
    rd += 36 * 256 * esz                                           # projection, fp32 feature rows out
    wr += 36 * 72 * 4
    return rd + wr


def h2d_rate_gbs(device, nbytes=1 << 30):
    """Pinned host -> device copy rate (GB/s), for the PCIe-inclusive figure; not part of the timed region."""
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    devb = torch.empty(nbytes, dtype=torch.uint8, device=device)
    devb.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        devb.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    return 3 * nbytes / (time.perf_counter() - t0) / 1e9


class Pipeline:
    """Device-resident synthetic batch + the per-step driver."""

    def __init__(self, n_frames, T, chunk_frames, seed, device, weight_seed=0, precision="fp32",
                 host_crop_cameras=False):
        self.dev = device
        self.host_crop_cameras = host_crop_cameras
        self.hm = load_generic_hand_model()
        self.model = model_from_state_dict(arch.make_synthetic_state(weight_seed)).to(device)
        self.model.precision = precision

        def fk(ja, wr, hidx):
            return skin_landmarks(self.hm, torch.from_numpy(ja), torch.from_numpy(wr),
                                  hand_idx=torch.from_numpy(hidx), mirror_right=True).cpu().numpy()

        wl = synthetic.make_workload(n_frames, T, seed, self.hm, fk)
        self.wl = wl
        d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)  # noqa: E731
        self.S = wl.joint_angles.shape[0]
        self.gt_angles, self.gt_wrist = d(wl.joint_angles, torch.float32), d(wl.wrist, torch.float32)
        self.hand_idx = d(wl.hand_idx, torch.int64)
        self.src_cams, self.dst_cams = d(wl.src_cams, torch.float64), d(wl.dst_cams, torch.float64)
        self.src_off, self.src_pitch = d(wl.src_offset, torch.int64), d(wl.src_pitch, torch.int32)
        self.K, self.E = d(wl.intrinsics, torch.float32), d(wl.extrinsics, torch.float32)
        # rig description for the on-device crop-camera generation (row A0): 4 fisheye cameras, no roll,
        # and the two views the workload assigns to each hand (section 8d forces exactly 2)
        self.rig_cams = d(wl.rig_cams, torch.float64)
        self.rig_angles = torch.zeros(synthetic.N_VIEWS, dtype=torch.float64, device=device)
        self.views = d(wl.views, torch.int32)
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.images = torch.randint(0, 256, (wl.image_bytes,), dtype=torch.uint8, device=device, generator=g)
        hm_m = scaled_hand_model(self.hm, 0.001)
        self.skel = InputSkeletonData(hm_m.joint_rotation_axes.to(device), hm_m.joint_rest_positions.to(device))
        self.hm_dev = hand_model_to(self.hm, device)     # constant model data resident like the weights: no per-call H2D copies
        # chunks of whole recordings; sample order is [rec][hand][t]
        assert chunk_frames % T == 0
        self.chunk = min(chunk_frames, n_frames) * 2           # samples per chunk
        cs = self.chunk
        self.sample_range = (torch.arange(cs, device=device)[:, None] * 2 + torch.tensor([0, 2], device=device)).contiguous()
        self.memory_idx = torch.arange(cs, device=device)
        self.use_memory = torch.zeros(cs, dtype=torch.bool, device=device)      # throughput mode, section 8(d)
        self.crops = torch.empty(2 * cs, 96, 96, dtype=torch.float32, device=device)
        self.pred_lm = torch.empty(wl.n_rec, 2, T, 21, 3, dtype=torch.float32, device=device)
        self.gt_lm = torch.empty_like(self.pred_lm)
        self.valid = torch.ones(wl.n_rec, 2, T, dtype=torch.bool, device=device)
        self.partials = None

    def step(self, marks=None):
        """One pass over the batch.  `marks`: optional list that receives (stage name, torch.cuda.Event) pairs recorded on
        the launch stream (= torch's current stream, which the library launches on) at every stage boundary."""
        def mark(name):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))
        cs = self.chunk
        pred = self.pred_lm.view(-1, 21, 3)
        gt = self.gt_lm.view(-1, 21, 3)
        mark("start")
        for s0 in range(0, self.S, cs):
            s1, c0, c1 = s0 + cs, 2 * s0, 2 * (s0 + cs)
            if self.host_crop_cameras:          # crop cameras precomputed on the host at set-up
                dst, K, E = self.dst_cams[c0:c1], self.K[c0:c1], self.E[c0:c1]
            else:                               # A0 on the device: FK of 3 poses -> look-at crop cameras -> K, E
                pts = gpu_crop.crop_points_batched(self.hm_dev, self.gt_angles[s0:s1], self.gt_wrist[s0:s1],
                                                   self.hand_idx[s0:s1], 63)
                cams = gpu_crop.gen_crop_cameras_batched(pts, self.rig_cams, self.rig_angles, self.hand_idx[s0:s1],
                                                         forced_views=self.views[s0:s1])
                dst, K, E = cams.dst_cams.view(-1, 20), cams.intrinsics.view(-1, 3, 3), cams.extrinsics.view(-1, 4, 4)
            mark("crop_cameras")
            gpu_warp.warp_crops(self.images, self.src_off[c0:c1], self.src_pitch[c0:c1], self.src_cams[c0:c1],
                                dst, out=self.crops)
            mark("warp")
            fd = InputFrameData(self.crops, K, E)
            desc = InputFrameDesc(self.sample_range, self.memory_idx, self.use_memory, self.hand_idx[s0:s1])
            out = self.model.regress_pose_use_skeleton(fd, desc, self.skel, required_rows=cs)
            mark("model")
            skin_landmarks(self.hm_dev, out.joint_angles, out.wrist_xfs, hand_idx=self.hand_idx[s0:s1], mirror_right=True,
                           wrist_translation_scale=1000.0, out=pred[s0:s1])
            skin_landmarks(self.hm_dev, self.gt_angles[s0:s1], self.gt_wrist[s0:s1], hand_idx=self.hand_idx[s0:s1],
                           mirror_right=True, out=gt[s0:s1])
            mark("fk")
        self.partials = mu.metric_partials(self.gt_lm, self.pred_lm, self.valid)
        mark("metric_partials")
        udist.all_reduce_partials(self.partials)
        return self.partials


class CpuOracle:
    """The CPU oracle (oracle/: a port of the reference path, pinned to its goldens) on samples of the same workload.
    Test infrastructure used as the reported CPU baseline and as the parity checker; never part of the product path."""

    def __init__(self, pipe):
        from oracle import fk as ofk, geometry as og, metrics as omx, model as om
        self.ofk, self.og, self.om = ofk, og, om
        self.pipe, self.wl = pipe, pipe.wl
        self.imgs = pipe.images.cpu().numpy()
        self.hm = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in pipe.hm._asdict().items() if v is not None}
        self.frame_bytes = synthetic.SRC_H * synthetic.SRC_W * synthetic.N_VIEWS
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        self.sd = om.to_torch_state(arch.make_synthetic_state(0))
        self.skel = SimpleNamespace(joint_rotation_axes=torch.from_numpy(self.hm["joint_rotation_axes"]).float(),
                                    joint_rest_positions=torch.from_numpy(self.hm["joint_rest_positions"]).float() * 0.001)

    def warp(self, h0, n):
        """Crops of hands [h0, h0+n): fisheye -> pinhole, cv2 fixed-point sampling, /255 (tracker.py:54-80, 210-211)."""
        wl, og = self.wl, self.og
        crops = np.empty((2 * n, 96, 96), np.float32)
        for i, k in enumerate(range(2 * h0, 2 * (h0 + n))):
            sc, dc = wl.src_cams[k], wl.dst_cams[k]
            frame = self.imgs[(wl.src_offset[k] // self.frame_bytes) * self.frame_bytes:][:self.frame_bytes].reshape(synthetic.SRC_H, -1)
            view = frame[:, (wl.src_offset[k] % self.frame_bytes):][:, :synthetic.SRC_W]
            src = og.Fisheye62(int(sc[0]), int(sc[1]), (sc[2], sc[3]), (sc[4], sc[5]), list(sc[6:14]), sc[14:].reshape(4, 4))
            dst = og.Pinhole(96, 96, (dc[0], dc[1]), (dc[2], dc[3]), dc[4:].reshape(4, 4))
            crops[i] = og.warp_crop(src, dst, np.ascontiguousarray(view))
        return crops

    def model(self, crops, h0, n):
        """One regress_pose_use_skeleton call on n hands, a fresh model like the reference's per-recording one."""
        wl = self.wl
        m = self.om.OracleModel(self.sd)
        fd = SimpleNamespace(left_images=torch.from_numpy(crops), intrinsics=torch.from_numpy(wl.intrinsics[2 * h0:2 * (h0 + n)]),
                             extrinsics_xf=torch.from_numpy(wl.extrinsics[2 * h0:2 * (h0 + n)]))
        desc = SimpleNamespace(sample_range=torch.arange(n)[:, None] * 2 + torch.tensor([0, 2]),
                               memory_idx=torch.arange(n), use_memory=torch.zeros(n, dtype=torch.bool),
                               hand_idx=torch.from_numpy(wl.hand_idx[h0:h0 + n]))
        with torch.no_grad():
            return m.regress_pose_use_skeleton(fd, desc, self.skel)

    def landmarks(self, out, h0, n):
        wl = self.wl
        hidx = wl.hand_idx[h0:h0 + n]
        wx = out["wrist_xfs"].numpy().astype(np.float64)
        wx[:, :3, 3] *= 1000.0
        wx[hidx == 1, :, 0] *= -1
        gtw = wl.wrist[h0:h0 + n].astype(np.float64).copy()
        gtw[hidx == 1, :, 0] *= -1
        pred = self.ofk.skin_landmarks(self.hm, out["joint_angles"].numpy(), wx)
        gt = self.ofk.skin_landmarks(self.hm, wl.joint_angles[h0:h0 + n], gtw)
        err = np.linalg.norm(gt.astype(np.float64) - pred.astype(np.float64), axis=-1).mean(-1)
        return pred, float(err.mean())

    def end_to_end(self, h0, n):
        return self.landmarks(self.model(self.warp(h0, n), h0, n), h0, n)


def cpu_baseline(pipe, n_hands, budget_s=12.0):
    """the CPU oracle timed model-only and end-to-end (warp + network + FK + error) at batch 2
    (the reference's per-frame call: one frame, two hands) and at batch `n_hands` (256), on the first hands of the
    same workload, all host cores torch uses.  The headline entry is end-to-end at the large batch.  Returns
    (cpu_baseline dict, oracle landmarks of the first n hands, their MPJPE)."""
    cpu = CpuOracle(pipe)
    n = min(n_hands, pipe.chunk)
    share = budget_s / 4.0

    def timed(fn, units, min_reps=1):
        t0 = time.perf_counter()
        reps, res = 0, None
        while reps < min_reps or (time.perf_counter() - t0) * (reps + 1) / max(reps, 1) < share:
            res = fn(reps)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return units / dt, dt, reps, res

    e2e_v, e2e_dt, e2e_reps, (pred, mpjpe) = timed(lambda r: cpu.end_to_end(0, n), n)
    crops = cpu.warp(0, n)
    mod_v, mod_dt, mod_reps, _ = timed(lambda r: cpu.model(crops, 0, n), n)
    nf = max(n // 2, 1)                                    # frames available for the per-frame configuration
    e2e2_v, e2e2_dt, e2e2_reps, _ = timed(lambda r: cpu.end_to_end(2 * (r % nf), 2), 2, min_reps=2)
    mod2_v, mod2_dt, mod2_reps, _ = timed(lambda r: cpu.model(crops[4 * (r % nf):4 * (r % nf) + 4], 2 * (r % nf), 2), 2, min_reps=2)
    cores = torch.get_num_threads()
    row = lambda v, dt, reps, what: {"value": v, "unit": "hand-frames/s", "s_per_call": dt, "calls": reps, "what": what}  # noqa: E731
    return {"value": e2e_v, "unit": "hand-frames/s", "cores": cores, "kind": "port",
            "sample": f"first {n} hand-frames ({2 * n} crops) of the same workload in one call, {e2e_reps} pass(es), "
                      f"oracle warp(cv2 fixed-point)+network fp32+FK, {e2e_dt:.2f} s/pass",
            "matrix": {f"end_to_end_batch{n}": row(e2e_v, e2e_dt, e2e_reps, "warp + network + FK + error, one call"),
                       f"model_only_batch{n}": row(mod_v, mod_dt, mod_reps, "network only (crops precomputed), one call"),
                       "end_to_end_batch2": row(e2e2_v, e2e2_dt, e2e2_reps, "the reference's per-frame call: 1 frame = 2 hands = 4 crops"),
                       "model_only_batch2": row(mod2_v, mod2_dt, mod2_reps, "network only, 2 hands per call"),
                       "note": "SURVEY.md section 8(d) matrix; every cell is the oracle port on the GPU box's host cores "
                               f"({cores} torch threads), bounded to about {budget_s:.0f} s in all"}}, pred, mpjpe


def per_frame_row(pipe, n_frames=256):
    """The reference's own call pattern (tracker.py:141-168, run_eval_known_skeleton.py:68-89): one
    HandTracker.track_frame per frame (1 frame = 2 hands = 4 crops, temporal memory on after the first frame) and
    landmarks_from_hand_pose per hand, on the first frames of the same workload; host images and camera objects as
    the stream reader delivers them.  The GPU row beside cpu_baseline.matrix.end_to_end_batch2."""
    from umetrack_amd.lib.common import camera
    from umetrack_amd.lib.tracker.perspective_crop import landmarks_from_hand_pose
    from umetrack_amd.lib.tracker.tracker import HandTracker, HandTrackerOpts, InputFrame, ViewData
    wl = pipe.wl
    T, W, H, V = wl.T, synthetic.SRC_W, synthetic.SRC_H, synthetic.N_VIEWS
    n_frames = min(n_frames, wl.n_rec * T)
    frames = pipe.images[:n_frames * W * H * V].cpu().numpy().reshape(n_frames, H, V, W)
    rig = [camera.Fisheye62CameraModel(W, H, (r[2], r[3]), (r[4], r[5]), list(r[6:14]), r[14:].reshape(4, 4).copy())
           for r in wl.rig_cams]
    work = []
    for f in range(n_frames):
        rec, t = divmod(f, T)
        crop_cams = {}
        for hand in range(2):
            s = (rec * 2 + hand) * T + t
            crop_cams[hand] = {int(wl.views[s, v]): camera.PinholePlaneCameraModel(
                96, 96, tuple(wl.dst_cams[2 * s + v, 0:2]), tuple(wl.dst_cams[2 * s + v, 2:4]), (),
                wl.dst_cams[2 * s + v, 4:].reshape(4, 4).copy()) for v in range(2)}
        views = [ViewData(image=frames[f, :, v, :], camera=rig[v], camera_angle=0.0) for v in range(V)]
        work.append((InputFrame(views=views), crop_cams))
    pipe.model.reset_memory()
    tracker = HandTracker(pipe.model, HandTrackerOpts())
    lm = np.zeros((n_frames, 2, 21, 3), np.float32)

    def run(lo, hi):
        for f in range(lo, hi):
            if f % T == 0:
                tracker.reset_history()            # a new recording starts without memory
            res = tracker.track_frame(work[f][0], pipe.hm, work[f][1])
            for hand, pose in res.hand_poses.items():
                lm[f, hand] = landmarks_from_hand_pose(pipe.hm, pose, hand)

    warm = min(8, n_frames)
    run(0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0, n_frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same frames through the bulk path's landmarks are not comparable (use_memory differs); check finiteness and
    # report the distance to ground truth instead
    gt = pipe.gt_lm.view(wl.n_rec, 2, T, 21, 3)[:(n_frames + T - 1) // T].permute(0, 2, 1, 3, 4).reshape(-1, 2, 21, 3)[:n_frames]
    return {"value": 2 * n_frames / dt, "unit": "hand-frames/s", "ms_per_frame": dt / n_frames * 1e3, "frames": n_frames,
            "hands_per_frame": 2, "crops_per_frame": 4, "finite": bool(np.isfinite(lm).all()),
            "mpjpe_vs_gt_mm": float(np.linalg.norm(lm - gt.cpu().numpy(), axis=-1).mean()),
            "host_syncs_per_frame": 1,
            "what": "HandTracker.track_frame + landmarks_from_hand_pose per hand, frame by frame, host uint8 views in, numpy "
                    "poses and landmarks out, temporal memory on after each recording's first frame; one pinned H2D copy, "
                    "warp + network + FK, one D2H copy and one stream synchronisation per frame (crop cameras given, as "
                    "in cpu_baseline.matrix.end_to_end_batch2)"}


def sustained_peaks():
    """What the matrix pipe of THIS GPU sustains right now, from the bare-loop probe tools/mfma_peak.hip (built by
    __graft_entry__.build() into tools/bin/): bf16 32x32x16 MFMAs in a 2x2-tile loop that re-reads both operands from
    LDS, two waves per SIMD - the loop shape of the backbone's convolution kernels - and fp32 32x32x2 on register
    operands.  Run as a child process after the timed region; None when the probe is not there."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "bin", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        txt = subprocess.run([exe, "brief"], capture_output=True, text=True, timeout=60, check=True).stdout
    except (subprocess.SubprocessError, OSError):
        return None
    got = {}
    for ln in txt.splitlines():
        m = re.search(r":\s+([0-9.]+) TFLOP/s", ln)
        if m and ln.startswith("bf16") and "LDS" in ln:
            got["bf16_lds_fed_tflops"] = float(m.group(1))
        elif m and ln.startswith("fp32"):
            got["fp32_tflops"] = float(m.group(1))
    if len(got) != 2:
        return None
    got["note"] = ("tools/mfma_peak.hip `brief`, run on this GPU after the timed region: bare MFMA loops over all 256 CUs, "
                   "2 waves per SIMD; bf16 = 2x2-tile loop feeding both operands from LDS (the convolution kernels' loop "
                   "shape), fp32 = register operands")
    return got


def stage_rows(stage_ms, pipe, backbone_ms, forward_ms, mfma_peak):
    """every stage of one step with its algorithmic bytes or FLOP, its HIP-event time and the
    fraction of the roofline that bounds it.  Times other than the head's are from one extra step after the timed
    region with an event at every stage boundary; the head is forward - backbone from the library's events inside
    the timed region."""
    S, n_crops = pipe.S, 2 * pipe.S
    rows = []

    def hbm(name, key, nbytes, per_unit):
        ms = stage_ms.get(key)
        if ms:
            gbs = nbytes / (ms * 1e-3) / 1e9
            rows.append({"stage": name, "bound": "hbm", "ms": ms, "algorithmic_bytes": nbytes, "per_unit": per_unit,
                         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})

    if not pipe.host_crop_cameras:
        hbm("crop cameras (FK of 3 poses + look-at + K/E packing)", "crop_cameras", S * (3 * 404 + 756 + 2 * 260),
            "3 x 404 B FK + 756 B in + 2 x 260 B out per hand; latency-bound fp64 4x4 algebra, HBM priced for completeness")
    hbm("warp (fisheye -> pinhole crops)", "warp", n_crops * 73728, "73 728 B per crop (9216 px x (4 B written + 4 x 1 B gathered))")
    head_ms = forward_ms - backbone_ms
    head_flop = (2.05e6 + 1.75e6 + 15.31e6) * S
    rows.append({"stage": "backbone (stem + 24 convolution launches)", "bound": "mfma", "ms": backbone_ms,
                 "algorithmic_flop": BACKBONE_FLOP_PER_CROP * n_crops, "per_unit": "969.23 MFLOP per crop",
                 "achieved": BACKBONE_FLOP_PER_CROP * n_crops / (backbone_ms * 1e-3) / 1e12, "peak": mfma_peak,
                 "unit": "TFLOP/s", "frac": BACKBONE_FLOP_PER_CROP * n_crops / (backbone_ms * 1e-3) / 1e12 / mfma_peak})
    rows.append({"stage": "head (fusion + temporal + regressor + decode)", "bound": "mfma", "ms": head_ms,
                 "algorithmic_flop": head_flop, "per_unit": "2.05 + 1.75 + 15.31 MFLOP per hand-frame",
                 "achieved": head_flop / (head_ms * 1e-3) / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": head_flop / (head_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS})
    hbm("FK (prediction + ground truth)", "fk", 2 * S * 404, "404 B per hand-frame (shared hand model), two launches")
    hbm("metric partials", "metric_partials", S * 504, "504 B per hand-frame")
    return rows


def stream_main(args, rank, world, dev):
    """configs[4]: one pass over the stream = the timed region; --steps / --warmup count whole passes."""
    from umetrack_amd.stream import StreamRunner
    runner = StreamRunner(dev, precision=args.precision, chunk_frames=args.frames, frames_per_recording=args.frames_per_recording)
    runner.run(min(args.stream_frames, args.frames * world), rank, world)          # warm-up: one chunk per rank
    for _ in range(max(args.warmup - 1, 0)):
        runner.run(args.stream_frames, rank, world)
    elapsed, res, lo, hi = 0.0, None, 0.0, 0.0
    cdev = dev if (world > 1 and torch.distributed.get_backend() == "nccl") else None
    for _ in range(args.steps):
        res = runner.run(args.stream_frames, rank, world)
        elapsed += udist.max_over_ranks(res.seconds, cdev)
        a, b = udist.min_max_over_ranks(res.seconds * 1e3, cdev)
        lo, hi = lo + a, hi + b
    collective = udist.describe(cdev)
    fin = mu.finalize_partials(res.partials)
    if rank != 0:
        return
    print(json.dumps({
        "metric": "hand-frames/sec (4-view)", "value": 2 * args.stream_frames * args.steps / elapsed, "unit": "hand-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32": "f32", "fp32x3": "f32 (3 x bf16 split operands, f32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[4]: synthetic stream of {args.stream_frames} frames x 2 hands x 2 views, produced "
                               f"on the device in chunks of {args.frames} frames, sharded by contiguous chunk over {world} GPU(s), "
                               "one all-reduce of the 108 float64 metric partials; the time includes generating the stream",
                   "stream_frames": args.stream_frames, "chunk_frames": args.frames,
                   "chunks_rank0": list(res.chunks_local), "weights": "seeded random (pretrained_weights.torch not distributed)"},
        "collective": dict(collective, ms_per_step_min_rank=lo / args.steps, ms_per_step_max_rank=hi / args.steps),
        "mpjpe_mm": fin.mpjpe, "mpjpa": fin.mpjpa, "n_valid": fin.n_valid, "n_total": fin.n_total}))


def progress(rank, msg):
    """One line on stderr per phase (every rank): a silent multi-minute run looks hung to a watchdog, and a real hang then
    names its phase."""
    print(f"[bench rank {rank} +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU (weak scaling)")
    ap.add_argument("--frames-per-recording", type=int, default=64)
    ap.add_argument("--chunk-frames", type=int, default=4096)
    ap.add_argument("--cpu-hands", type=int, default=256,
                    help="hand-frames in the large-batch cells of the CPU-baseline matrix (0 = skip the baseline)")
    ap.add_argument("--precision", choices=("fp32", "bf16", "fp32x3"), default="fp32",
                    help="backbone arithmetic; fp32 (fp32 matrix instructions, an fmaf chain per output) is the reference's and the "
                         "headline line; bf16 is a separate reduced-precision mode; fp32x3 keeps fp32 tensors and multiplies "
                         "three-way bf16 splits of the operands on the bf16 matrix pipe (fp32-level accuracy, reported separately)")
    ap.add_argument("--no-x3", action="store_true", help="skip the fp32x3 and bf16 side measurements of the default fp32 run")
    ap.add_argument("--host-crop-cameras", action="store_true",
                    help="use crop cameras precomputed on the host instead of generating them on the GPU each step")
    ap.add_argument("--stream-frames", type=int, default=0,
                    help="run BASELINE.json configs[4] instead: a synthetic stream of this many frames, produced on the "
                         "device in chunks of --frames, sharded over the ranks by contiguous chunk (STRONG scaling)")
    ap.add_argument("--per-frame-frames", type=int, default=256,
                    help="frames of the per-frame tracker row (HandTracker.track_frame, 2 hands x 2 views per call; 0 = skip)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host->device copy-rate measurement")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: all ranks share cuda:0 (use with --backend gloo on a 1-GPU box)")
    args = ap.parse_args()

    rank, world, local = udist.init_from_env(args.backend)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.stream_frames > 0:
        return stream_main(args, rank, world, dev)

    progress(rank, f"world {world}, device {dev}: building the synthetic workload")
    pipe = Pipeline(args.frames, args.frames_per_recording, args.chunk_frames, seed=rank, device=dev,
                    precision=args.precision, host_crop_cameras=args.host_crop_cameras)
    progress(rank, f"{args.warmup} warm-up step(s)")
    for _ in range(args.warmup):
        pipe.step()
    torch.cuda.synchronize()
    udist.barrier()
    progress(rank, f"timing {args.steps} step(s)")
    pipe.model.timing_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    torch.cuda.synchronize()
    udist.barrier()
    local_elapsed = time.perf_counter() - t0
    cdev = dev if (world > 1 and torch.distributed.get_backend() == "nccl") else None    # gloo reduces host tensors
    elapsed = udist.max_over_ranks(local_elapsed, cdev)
    rank_ms = udist.min_max_over_ranks(local_elapsed / args.steps * 1e3, cdev)
    collective = udist.describe(cdev)
    progress(rank, f"timed region done: {elapsed / args.steps * 1e3:.2f} ms per step (max over ranks)")
    bb_ms, fwd_ms, calls = pipe.model.timing_read()
    pipe.model.timing_enable(False)
    fin = mu.finalize_partials(pipe.partials)
    parity_pred = pipe.pred_lm.view(-1, 21, 3)[:max(args.cpu_hands, 1)].cpu().numpy()   # throughput-mode landmarks

    # second mode of section 8(d): the same batch as 64-frame recordings in lock-step with the temporal
    # memory ON (bulk backbone, then T sequential memory+regressor steps); per-rank, no collective
    from umetrack_amd import bulk
    runner = bulk.BulkSequenceRunner(pipe.model, pipe.wl.n_rec, pipe.wl.T, chunk_samples=pipe.chunk)
    rec_valid = pipe.valid.view(-1, pipe.wl.T)

    def recurrent_step():
        runner.features(pipe.images, pipe.src_off, pipe.src_pitch, pipe.src_cams, pipe.dst_cams, pipe.K, pipe.E)
        out = runner.run(pipe.E, pipe.hand_idx, pipe.skel, rec_valid)
        skin_landmarks(pipe.hm, out.joint_angles, out.wrist_xfs, hand_idx=pipe.hand_idx, mirror_right=True,
                       wrist_translation_scale=1000.0, out=pipe.pred_lm.view(-1, 21, 3))
        return mu.metric_partials(pipe.gt_lm, pipe.pred_lm, pipe.valid)

    def time_recurrent():
        recurrent_step()
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        partials = recurrent_step()
        torch.cuda.synchronize()
        return time.perf_counter() - tr0, mu.finalize_partials(partials)

    # side measurements (rank 0's GPU only, no collective in them; never `value`): the same step in the two other
    # backbone modes, 1 warm-up + 2 timed steps each, backbone time from the library's HIP events
    def side_mode(precision, peak, note):
        progress(rank, f"side measurement: {precision}")
        pipe.model.precision = precision
        pipe.step()
        torch.cuda.synchronize()
        pipe.model.timing_enable(True)
        tx = time.perf_counter()
        for _ in range(2):
            pipe.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - tx) / 2
        bb, fwd, ncalls = pipe.model.timing_read()
        pipe.model.timing_enable(False)
        pred_side = pipe.pred_lm.clone()
        rec_side_dt, _ = time_recurrent()
        pipe.pred_lm.copy_(pred_side)
        pipe.model.precision = args.precision
        tf = BACKBONE_FLOP_PER_CROP * 2 * pipe.chunk / (bb / ncalls * 1e-3) / 1e12
        diff = (pipe.pred_lm - pred32).abs().reshape(-1)
        return {"value": pipe.S / dt, "unit": "hand-frames/s", "ms_per_step": dt * 1e3, "backbone_ms": bb / ncalls,
                "forward_ms": fwd / ncalls, "backbone_tflops": tf, "peak_tflops": peak, "frac": tf / peak,
                "recurrent_value": pipe.S / rec_side_dt, "recurrent_ms_per_step": rec_side_dt * 1e3,
                "max_landmark_diff_vs_fp32_mm": float(diff.max()),
                "p999_landmark_diff_vs_fp32_mm": float(torch.quantile(diff[:: max(1, diff.numel() // 4_000_000)].double(), 0.999)),
                "mean_landmark_diff_vs_fp32_mm": float(diff.mean()),
                "pred": pipe.pred_lm.view(-1, 21, 3)[:max(args.cpu_hands, 1)].cpu().numpy(), "note": note}

    x3 = bf16_side = None
    if args.precision == "fp32" and not args.no_x3 and world == 1:
        pred32 = pipe.pred_lm.clone()
        x3 = side_mode("fp32x3", BF16_MATRIX_PEAK_TFLOPS / 6,
                       "UMT_PREC_FP32_BF16X3: fp32 tensors; each fp32 operand of the stride-1 3x3 backbone layers is the exact "
                       "sum of three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulation; peak = bf16 "
                       "dense peak / 6; 2 timed steps")
        bf16_side = side_mode("bf16", BF16_MATRIX_PEAK_TFLOPS,
                              "UMT_PREC_BF16: bf16 backbone activations and weights, fp32 accumulation; fusion, temporal memory, "
                              "head and FK stay fp32 tensors (the head's 3x3 layers on the split-operand kernels of the fp32x3 mode for batches "
                              "of 228 hands or more: fp32-rounding-level accuracy); peak = bf16 dense matrix peak; 2 timed steps; the difference from the fp32 "
                              "mode depends on the weights: `stated_tolerance_mm` is 1.5 x the worst of 8 weight seeds on this "
                              "batch (profiles/r03_bf16_tolerance_8seeds.json, tools/bf16_tolerance.py; held by "
                              "tests/test_gpu_bf16.py); this run's weight seed is 0")
        bf16_side["stated_tolerance_mm"] = dict(UmeTrackModel.BF16_LANDMARK_TOLERANCE_MM)
        bf16_side["tolerance_source"] = "profiles/r03_bf16_tolerance_8seeds.json"

    # per-launch view of the backbone (outside the timed region: the library records a HIP event between every two
    # launches, which costs a little): one more step in the headline precision
    progress(rank, "per-launch and per-stage steps, recurrent mode")
    pipe.model.layer_timing_enable(True)
    pipe.step()
    layer_rows = pipe.model.layer_timing_read()
    pipe.model.layer_timing_enable(False)

    # per-stage view (section 8d): one more step with a HIP event on the launch stream at every stage boundary
    marks = []
    pipe.step(marks)
    torch.cuda.synchronize()
    stage_ms = {}
    for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
        stage_ms[name] = stage_ms.get(name, 0.0) + e0.elapsed_time(e1)

    rec_dt, rec_fin = time_recurrent()
    progress(rank, "per-frame tracker path")
    per_frame = per_frame_row(pipe, args.per_frame_frames) if args.per_frame_frames > 0 and rank == 0 else None

    hands_per_step = pipe.S * world
    if rank != 0:
        return
    crops_per_call = 2 * pipe.chunk
    bf16 = args.precision == "bf16"
    traffic = traffic_head = None
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_backbone_traffic.json") for r in (3, 2, 1))
                  if os.path.exists(q)), "")
    if tpath:      # PMC passes cannot run inside this process; see the file for provenance
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic_head = tj.get("measured_at_head")
        tj = tj.get(args.precision, tj if not bf16 else {})
        if "read_bytes_per_crop" in tj:
            traffic = (tj["read_bytes_per_crop"] + tj["write_bytes_per_crop"]) * crops_per_call
    achieved = BACKBONE_FLOP_PER_CROP * crops_per_call / (bb_ms / calls * 1e-3) / 1e12
    peak = BF16_MATRIX_PEAK_TFLOPS if bf16 else FP32_MATRIX_PEAK_TFLOPS
    if args.precision == "fp32x3":       # six bf16 matrix instructions per fp32 product
        peak = BF16_MATRIX_PEAK_TFLOPS / 6
    progress(rank, "matrix-pipe probe (sustained peak of this GPU)")
    sus = sustained_peaks()

    def sustained_of(precision):
        if sus is None:
            return None
        return {"fp32": sus["fp32_tflops"], "bf16": sus["bf16_lds_fed_tflops"], "fp32x3": sus["bf16_lds_fed_tflops"] / 6}[precision]
    peak_note = {"fp32": "fp32 matrix peak (v_mfma_f32_32x32x2_f32)", "bf16": "bf16 dense matrix peak",
                 "fp32x3": "bf16 dense matrix peak / 6: six v_mfma_f32_32x32x16_bf16 per fp32 product; achieved counts each "
                           f"fp32 multiply-add once (= {achieved / FP32_MATRIX_PEAK_TFLOPS:.3f} of the fp32 matrix peak)"}[args.precision]
    act_bytes = backbone_activation_bytes_per_crop(2 if bf16 else 4) * crops_per_call
    hbm_achieved = act_bytes / (bb_ms / calls * 1e-3) / 1e9
    # the dominant kernel: the stride-1 3x3 layers with 128 and 256 channels all run one template instance (with and
    # without residual); its launches are rows of the rocprofv3 --kernel-trace --stats summary under profiles/
    dom = [r for r in layer_rows if r[0].split()[0] in ("c1", "c2+res") and int(r[0].split()[1].split("->")[1]) >= 128]
    dom_kernel = {"fp32": "conv3x3s1_halo_f32<2,2,2,2,RES>", "bf16": "conv3x3s1_halo_bf16<2,2,2,2,RES>",
                  "fp32x3": "conv3x3s1_halo_x3<2,2,2,2,RES>"}[args.precision]
    dom_launches = sum(r[3] for r in dom)
    dom_ms = sum(r[2] for r in dom) / max(dom_launches, 1)
    dom_flop = sum(r[1] for r in dom) / max(dom_launches, 1)
    line = {
        "metric": "hand-frames/sec (4-view)", "value": hands_per_step * args.steps / elapsed, "unit": "hand-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32": "f32", "fp32x3": "f32 (3 x bf16 split operands, f32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[2]: synthetic 4-view x 2-hand batch, {args.frames} frames per GPU, "
                               f"2 crops per hand, {args.precision} MFMA backbone, known-skeleton head, use_memory=False",
                   "frames_per_gpu": args.frames, "hand_frames_per_gpu": pipe.S, "crops_per_gpu": 2 * pipe.S,
                   "source_image": f"{synthetic.SRC_W}x{synthetic.SRC_H} uint8 x {synthetic.N_VIEWS} views",
                   "chunk_frames": args.chunk_frames, "weights": "seeded random (pretrained_weights.torch not distributed)",
                   "parallelism": f"frames sharded over {world} GPU(s); all-reduce of 108 float64 metric partials"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "peak_note": peak_note,
                     "sustained_peak": sustained_of(args.precision),
                     "frac_vs_sustained": achieved / sustained_of(args.precision) if sus else None,
                     "sustained_probe": sus, "traffic": traffic,
                     "traffic_measured_at_head": traffic_head,
                     "traffic_note": "HBM bytes per forward call from rocprofv3 FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE, "
                                     f"profiles/{os.path.basename(tpath)} (separate --pmc passes of tools/round_profiles.sh; r02 on: measured at "
                                     "16384 crops per forward call, the size of this run's calls)",
                     "hbm_view": {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": hbm_achieved / HBM_PEAK_GBS,
                                  "note": "same launches priced by bytes: unfused input+output(+residual) rows of every "
                                          "backbone launch; the bf16 backbone sits between the two rooflines"},
                     "kernel": "backbone: stem + 24 " + {"bf16": "conv_*_bf16", "fp32": "conv_igemm_f32 / conv3x3s1_halo_f32 / conv3x3s2p_ds_f32 / conv3x3_pipe32_f32",
                                                          "fp32x3": "conv3x3s1_halo_x3 / conv3x3s2_ds_x3 (+ conv3x3s2p_ds_f32, conv_igemm_f32)"}[args.precision]
                               + " launches per forward call",
                     "flop_per_launch": BACKBONE_FLOP_PER_CROP * crops_per_call, "avg_ms": bb_ms / calls,
                     "forward_avg_ms": fwd_ms / calls, "calls": calls,
                     "dominant_kernel": {"kernel": dom_kernel, "launches_per_call": dom_launches,
                                         "share_of_backbone_time": sum(r[2] for r in dom) / max(sum(r[2] for r in layer_rows), 1e-9),
                                         "flop_per_launch": dom_flop, "avg_ms": dom_ms,
                                         "achieved": dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else None,
                                         "frac": dom_flop / (dom_ms * 1e-3) / 1e12 / peak if dom_ms > 0 else None,
                                         "note": "HIP events recorded by the library between the launches of one extra step after "
                                                 "the timed region; stride-1 3x3 layers with 128 and 256 channels"},
                     "layers": [{"launch": r[0], "ms": r[2] / max(r[3], 1), "tflops": r[1] / max(r[2], 1e-9) / 1e9} for r in layer_rows]},
        "stages": stage_rows(stage_ms, pipe, bb_ms / calls, fwd_ms / calls, peak),
        "model_tflops": MODEL_FLOP_PER_HAND * pipe.S * args.steps / elapsed / 1e12,
        "mpjpe_mm": fin.mpjpe, "mpjpa": fin.mpjpa,
        "collective": dict(collective, ms_per_step_min_rank=rank_ms[0], ms_per_step_max_rank=rank_ms[1],
                           note="all-reduce(sum) of the 108 float64 metric partials, once per step; allreduce_us = mean of "
                                "20 timed calls (slowest rank); ms_per_step_* = each rank's own clock over the timed region"),
        "recurrent": {"value_per_gpu": pipe.S / rec_dt, "unit": "hand-frames/s", "ms_per_step": rec_dt * 1e3,
                      "note": f"{pipe.wl.n_rec} recordings x {pipe.wl.T} frames in lock-step, use_memory=True after "
                              "frame 0, rank 0 only, 1 timed pass", "mpjpe_mm": rec_fin.mpjpe},
    }
    if per_frame is not None:
        line["per_frame"] = per_frame
    progress(rank, "host-to-device copy measurements")
    if not args.no_pcie:
        rate = h2d_rate_gbs(dev)
        frame_bytes = pipe.wl.image_bytes
        t_step = elapsed / args.steps
        line["pcie_inclusive"] = {"value": pipe.S / (t_step + frame_bytes / (rate * 1e9)), "unit": "hand-frames/s per GPU",
                                  "h2d_GBs": rate, "frame_bytes_per_step": frame_bytes,
                                  "note": "step time plus an un-overlapped pinned host->device copy of the step's source "
                                          "frames at the measured rate; `value` above has the frames resident in HBM"}
    if not args.no_pcie and world == 1:     # (single process only: pipe.step() ends in the all-reduce, and the other ranks are gone)
        # measured, not estimated: the next step's frames are copied from pinned host memory on a second stream into a
        # second device buffer while this step computes (what a reader feeding the GPU over PCIe would do)
        ov_steps = 3
        host = torch.empty(frame_bytes, dtype=torch.uint8).pin_memory()
        host.copy_(pipe.images)
        bufs = [pipe.images, torch.empty_like(pipe.images)]
        copy_stream = torch.cuda.Stream(device=dev)
        main_stream = torch.cuda.current_stream(dev)
        bufs[1].copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        to = time.perf_counter()
        for i in range(ov_steps):
            cur, nxt = bufs[(i + 1) % 2], bufs[i % 2]
            copy_stream.wait_stream(main_stream)            # the step that last read `nxt` has been enqueued before
            with torch.cuda.stream(copy_stream):
                nxt.copy_(host, non_blocking=True)
            pipe.images = cur
            pipe.step()
            main_stream.wait_stream(copy_stream)
        torch.cuda.synchronize()
        ov_dt = (time.perf_counter() - to) / ov_steps
        pipe.images = bufs[0]
        line["pcie_inclusive"]["overlapped"] = {
            "value": pipe.S / ov_dt, "unit": "hand-frames/s per GPU", "ms_per_step": ov_dt * 1e3, "steps": ov_steps,
            "note": "measured: every step's source frames arrive over PCIe from pinned host memory, double-buffered on a "
                    "copy stream beside the previous step's kernels; bound by max(step, copy)"}
        del host, bufs
    if args.cpu_hands > 0:
        progress(rank, f"CPU oracle baseline on {args.cpu_hands} hands (about 12 s)")
        base, cpu_pred, cpu_mpjpe = cpu_baseline(pipe, args.cpu_hands)
        line["cpu_baseline"] = base
        n = cpu_pred.shape[0]
        line["parity_sample_max_mm"] = float(np.abs(parity_pred[:n] - cpu_pred).max())
        for side in (x3, bf16_side):
            if side is not None:
                side["parity_sample_max_mm"] = float(np.abs(side["pred"][:n] - cpu_pred).max())
    for key, side in (("fp32_bf16x3", x3), ("bf16", bf16_side)):
        if side is not None:
            side.pop("pred")
            if sus:
                side["sustained_peak_tflops"] = sustained_of("fp32x3" if key == "fp32_bf16x3" else "bf16")
                side["frac_vs_sustained"] = side["backbone_tflops"] / side["sustained_peak_tflops"]
            line[key] = side
    print(json.dumps(line))


if __name__ == "__main__":
    main()
