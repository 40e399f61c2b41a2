"""BASELINE.json configs[4] / SURVEY.md section 8(d) "Stream": a long synthetic 4-view x 2-hand stream, produced on
the device in chunks of 4096 frames, sharded over the ranks by contiguous chunk, with ONE all-reduce of the
108-word MPJPE / MPJPA / PCK partial vector at the end (RCCL over xGMI on GPUs).

Every chunk is a function of its index only (poses, wrist transforms and source frames come from a device
generator seeded with base_seed + chunk index), so the stream - and therefore the reduced metrics up to float64
summation order - does not depend on how many ranks share it.  A chunk holds whole 64-frame recordings, so
MPJPA (three consecutive frames of one recording, load_eval.py:29-37) needs no halo between chunks.

The per-chunk step is the one bench.py times: crop cameras from the hand poses on the device (row A0/F1) ->
warp -> network -> forward kinematics of prediction and ground truth -> metric partials.  Crops the reference
would refuse (ValueError "Unable to create crop camera") are counted and excluded through the valid mask, like
tracking failures (tracker.py:141-168).
"""
import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from . import arch, dist as udist, synthetic
from . import _native as nat


@dataclass
class StreamResult:
    partials: torch.Tensor          # float64[108], already summed over ranks
    frames_total: int               # frames of the whole stream
    frames_local: int               # frames this rank processed
    chunks_local: Tuple[int, int]   # [first, last) chunk of this rank
    invalid_local: int              # hand-frames excluded because a crop camera was degenerate
    seconds: float                  # wall time of this rank's loop (synchronised)


def chunk_plan(total_frames: int, chunk_frames: int, frames_per_recording: int):
    """-> (number of chunks, frames in the last chunk).  Whole recordings only."""
    if chunk_frames % frames_per_recording or total_frames % frames_per_recording:
        raise ValueError("stream and chunk lengths must be whole recordings")
    n_chunks = (total_frames + chunk_frames - 1) // chunk_frames
    return n_chunks, total_frames - (n_chunks - 1) * chunk_frames


def run_sharded(n_chunks: int, rank: int, world: int, run_chunk: Callable[[int], torch.Tensor],
                zero: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """Contiguous chunk range of `rank`, partials accumulated locally, one all-reduce at the end."""
    c0, c1 = udist.shard_range(n_chunks, rank, world)
    acc = zero.clone()
    for c in range(c0, c1):
        acc += run_chunk(c)
    udist.all_reduce_partials(acc)
    return acc, (c0, c1)


class StreamRunner:
    """Device-side generator + per-chunk driver.  One instance per rank."""

    def __init__(self, device: torch.device, precision: str = "fp32", chunk_frames: int = 4096,
                 frames_per_recording: int = 64, base_seed: int = 1000, weight_seed: int = 0, rig_seed: int = 0):
        from .lib.common.hand import hand_model_to, load_generic_hand_model, scaled_hand_model
        from .lib.models.model_loader import model_from_state_dict
        from .lib.models.umetrack_model import InputSkeletonData
        from .lib.tracker import warp as gpu_warp
        self.dev = nat.require_gpu(device)
        self.T = frames_per_recording
        self.chunk_frames = chunk_frames
        self.base_seed = base_seed
        self.hm = load_generic_hand_model()
        self.hm_dev = hand_model_to(self.hm, self.dev)     # constant model data resident on the GPU
        self.model = model_from_state_dict(arch.make_synthetic_state(weight_seed)).to(self.dev)
        self.model.precision = precision
        hm_m = scaled_hand_model(self.hm, 0.001)
        self.skel = InputSkeletonData(hm_m.joint_rotation_axes.to(self.dev), hm_m.joint_rest_positions.to(self.dev))
        intr, coeffs, rig_c2w = synthetic.make_rig(np.random.default_rng(rig_seed))
        rig = np.stack([gpu_warp.pack_fisheye(synthetic.SRC_W, synthetic.SRC_H, intr[v, :2], intr[v, 2:], coeffs[v], rig_c2w[v])
                        for v in range(synthetic.N_VIEWS)])
        self.rig_cams = torch.from_numpy(rig).to(self.dev)
        self.rig_angles = torch.zeros(synthetic.N_VIEWS, dtype=torch.float64, device=self.dev)
        self.rig_pos = torch.from_numpy(rig_c2w[:, :3, 3].copy()).to(self.dev, torch.float32)
        self.rig_axis = torch.from_numpy(rig_c2w[:, :3, 2].copy()).to(self.dev, torch.float32)
        lim = torch.as_tensor(np.asarray(self.hm.joint_limits, dtype=np.float32), device=self.dev)
        self.lim_lo, self.lim_hi = lim[:, 0].contiguous(), lim[:, 1].contiguous()
        self._layout: Dict[int, dict] = {}
        self.invalid = 0

    # ---- fixed index tensors of a chunk with n_frames frames
    def _chunk_layout(self, n_frames: int) -> dict:
        lay = self._layout.get(n_frames)
        if lay is None:
            dev, T = self.dev, self.T
            n_rec = n_frames // T
            S = n_rec * 2 * T
            rec = torch.arange(n_rec, device=dev).repeat_interleave(2 * T)
            hand = torch.arange(2, device=dev).repeat_interleave(T).repeat(n_rec)
            t = torch.arange(T, device=dev).repeat(2 * n_rec)
            frame_bytes = synthetic.SRC_H * synthetic.SRC_W * synthetic.N_VIEWS
            lay = dict(n_rec=n_rec, S=S, hand=hand.to(torch.int64), frame_off=((rec * T + t) * frame_bytes).to(torch.int64),
                       image_bytes=n_frames * frame_bytes,
                       sample_range=(torch.arange(S, device=dev)[:, None] * 2 + torch.tensor([0, 2], device=dev)).contiguous(),
                       memory_idx=torch.arange(S, device=dev), use_memory=torch.zeros(S, dtype=torch.bool, device=dev),
                       pitch=torch.full((2 * S,), synthetic.SRC_W * synthetic.N_VIEWS, dtype=torch.int32, device=dev))
            self._layout[n_frames] = lay
        return lay

    def generate(self, chunk: int, n_frames: int):
        """Poses and source frames of chunk `chunk`: the distributions of synthetic.make_workload, drawn on the device."""
        lay = self._chunk_layout(n_frames)
        S, dev = lay["S"], self.dev
        g = torch.Generator(device=dev)
        g.manual_seed(self.base_seed + chunk)
        ja = self.lim_lo + torch.rand(S, 22, device=dev, generator=g) * (self.lim_hi - self.lim_lo)
        ja[:, 20:] = 0
        q = torch.randn(S, 4, device=dev, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        w, x, y, z = q.unbind(1)
        R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                         torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                         torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
        u = torch.rand(S, 3, device=dev, generator=g)
        wrist = torch.eye(4, device=dev).repeat(S, 1, 1)
        wrist[:, :3, :3] = R
        wrist[:, 0, 3] = -120 + 240 * u[:, 0]
        wrist[:, 1, 3] = -90 + 180 * u[:, 1]
        wrist[:, 2, 3] = 350 + 150 * u[:, 2]
        images = torch.randint(0, 256, (lay["image_bytes"],), dtype=torch.uint8, device=dev, generator=g)
        return ja.contiguous(), wrist.contiguous(), images

    def run_chunk(self, chunk: int, n_frames: Optional[int] = None) -> torch.Tensor:
        """-> float64[108] metric partials of the chunk (on the device)."""
        from .lib.common import metric_utils as mu
        from .lib.common.hand_skinning import skin_landmarks
        from .lib.models.umetrack_model import InputFrameData, InputFrameDesc
        from .lib.tracker import crop_cameras as gpu_crop
        from .lib.tracker import warp as gpu_warp
        n_frames = self.chunk_frames if n_frames is None else n_frames
        lay = self._chunk_layout(n_frames)
        S, hand = lay["S"], lay["hand"]
        ja, wrist, images = self.generate(chunk, n_frames)
        pts = gpu_crop.crop_points_batched(self.hm_dev, ja, wrist, hand, 63)
        # the two rig cameras whose optical axis is closest to the hand, in index order (section 8d: exactly 2 views)
        rel = pts[:, :21].mean(dim=1)[:, None, :] - self.rig_pos[None]
        cosang = (rel / rel.norm(dim=2, keepdim=True) * self.rig_axis[None]).sum(-1)
        views = cosang.topk(2, dim=1).indices.sort(dim=1).values.to(torch.int32).contiguous()
        cams = gpu_crop.gen_crop_cameras_batched(pts, self.rig_cams, self.rig_angles, hand, forced_views=views)
        crop_view = views.reshape(-1).to(torch.int64)
        src_off = lay["frame_off"].repeat_interleave(2) + crop_view * synthetic.SRC_W
        crops = gpu_warp.warp_crops(images, src_off, lay["pitch"], self.rig_cams[crop_view], cams.dst_cams.view(-1, 20))
        fd = InputFrameData(crops, cams.intrinsics.view(-1, 3, 3), cams.extrinsics.view(-1, 4, 4))
        desc = InputFrameDesc(lay["sample_range"], lay["memory_idx"], lay["use_memory"], hand)
        out = self.model.regress_pose_use_skeleton(fd, desc, self.skel, required_rows=S)
        pred = skin_landmarks(self.hm_dev, out.joint_angles, out.wrist_xfs, hand_idx=hand, mirror_right=True,
                              wrist_translation_scale=1000.0)
        gt = pts[:, :21].contiguous()                      # the tracked pose's landmarks are the ground truth
        ok = (cams.status == 0).all(dim=1)
        self.invalid += int((~ok).sum().item())
        shape = (lay["n_rec"], 2, self.T)
        return mu.metric_partials(gt.view(*shape, 21, 3), pred.view(*shape, 21, 3), ok.view(*shape))

    def run(self, total_frames: int, rank: int = 0, world: int = 1) -> StreamResult:
        n_chunks, last = chunk_plan(total_frames, self.chunk_frames, self.T)
        zero = torch.zeros(nat.METRIC_WORDS, dtype=torch.float64, device=self.dev)
        self.invalid = 0
        frames = [0]

        def one(c):
            n = last if c == n_chunks - 1 else self.chunk_frames
            frames[0] += n
            return self.run_chunk(c, n)

        torch.cuda.synchronize(self.dev)
        udist.barrier()
        t0 = time.perf_counter()
        acc, rng = run_sharded(n_chunks, rank, world, one, zero)
        torch.cuda.synchronize(self.dev)
        dt = time.perf_counter() - t0
        return StreamResult(acc, total_frames, frames[0], rng, self.invalid, dt)
