"""Synthetic 4-view x 2-hand workload of SURVEY.md section 8(d) / BASELINE.json configs[2].

Used by bench.py, __graft_entry__.smoke() and tests.  Everything here is host-side numpy
(row A0 geometry, batched) plus buffer packing; the arithmetic of the hot path itself
stays in the HIP library.

Sample order is [recording][hand][frame-in-recording], two crops per sample, so that
landmarks come out directly in the [n_rec, 2, T, 21, 3] layout of the metric kernel
(the per-recording arrays of run_eval_known_skeleton.py:64-66).
"""
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from .lib.common.affine import rot_z_deg
from .lib.tracker import warp as gpu_warp

SRC_W, SRC_H = 640, 480          # not stated by the reference (read from label JSON); section 8(d) picks 640x480
N_VIEWS = 4
CROP = 96


def make_rig(rng) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """4 Fisheye62 cameras: intrinsics row (fx,fy,cx,cy), 8 coefficients, camera_to_world (rig frame, mm)."""
    intr = np.tile(np.array([270.0, 270.0, (SRC_W - 1) / 2, (SRC_H - 1) / 2]), (N_VIEWS, 1))
    coeffs = np.concatenate([rng.uniform(-0.02, 0.02, (N_VIEWS, 4)), rng.uniform(-1e-3, 1e-3, (N_VIEWS, 2)),
                             rng.uniform(-0.02, 0.02, (N_VIEWS, 2))], axis=1)
    c2w = np.tile(np.eye(4), (N_VIEWS, 1, 1))
    for v, (yaw, pitch, tx, ty) in enumerate(((-30, 15, -30, 10), (30, 15, 30, 10), (-30, -15, -30, -10), (30, -15, 30, -10))):
        cy, sy = np.cos(np.deg2rad(yaw)), np.sin(np.deg2rad(yaw))
        cp, sp = np.cos(np.deg2rad(pitch)), np.sin(np.deg2rad(pitch))
        ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        c2w[v, :3, :3] = ry @ rx
        c2w[v, :3, 3] = (tx, ty, 0.0)
    return intr, coeffs, c2w


def random_rotations(rng, n) -> np.ndarray:
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def crop_cameras_batched(cam_c2w: np.ndarray, cam_angle: np.ndarray, pts_world: np.ndarray, mirror: np.ndarray,
                         focal_multiplier: float = 0.95, size: int = CROP):
    """Vectorised gen_crop_parameters_from_points (lib/common/crop.py:31-82, affine.py:34-66).

    cam_c2w [B,4,4] source camera per crop, cam_angle [B] degrees, pts_world [B,P,3] (float32 like
    the reference's landmarks), mirror [B] bool.  Returns f [B], c (scalar), crop cam_to_world
    [B,4,4], ok [B] (False where the reference would raise ValueError).
    """
    B = pts_world.shape[0]
    w2e = np.linalg.inv(cam_c2w)
    centre = ((pts_world.min(axis=1) + pts_world.max(axis=1)) / 2.0).astype(np.float64)   # float32 rounding kept
    local = np.einsum("bij,bj->bi", w2e[:, :3, :3], centre) + w2e[:, :3, 3]
    zdir = local / np.linalg.norm(local, axis=1, keepdims=True)
    bn = zdir / np.maximum(5.43e-20, np.linalg.norm(zdir, axis=1, keepdims=True))
    ax = np.stack([-bn[:, 1], bn[:, 0], np.zeros(B)], -1)                     # (0,0,1) x b
    K = np.zeros((B, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -ax[:, 2], ax[:, 1]
    K[:, 1, 0], K[:, 1, 2] = ax[:, 2], -ax[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -ax[:, 1], ax[:, 0]
    s2 = (ax * ax).sum(-1)
    delta = np.eye(3) + K + (K @ K) * ((1 - bn[:, 2]) / np.maximum(s2, 1e-15))[:, None, None]
    e2w = np.linalg.inv(w2e)
    new = e2w.copy()
    rz = np.stack([rot_z_deg(a) for a in cam_angle]) if np.any(cam_angle != 0) else np.tile(np.eye(3), (B, 1, 1))
    new[:, :3, :3] = (e2w[:, :3, :3] @ delta) @ rz
    new_w2e = np.linalg.inv(new)
    flip = np.where(mirror, -1.0, 1.0)
    new_w2e[:, 0, :] *= flip[:, None]
    pts_eye = np.einsum("bij,bpj->bpi", new_w2e[:, :3, :3], pts_world.astype(np.float64)) + new_w2e[:, None, :3, 3]
    ndc = pts_eye[..., :2] / pts_eye[..., 2:]
    c = (size - 1) / 2
    f = c / np.abs(ndc).reshape(B, -1).max(axis=1)
    ok = ~((pts_eye[..., 2] < 0.0001).any(axis=1) | (f < 5))
    return focal_multiplier * f, c, np.linalg.inv(new_w2e), ok


@dataclass
class Workload:
    """Host description of one batch; device tensors are made by bench.py / smoke()."""
    n_rec: int
    T: int
    joint_angles: np.ndarray      # [S,22] float32, S = n_rec*2*T samples in [rec][hand][t] order
    wrist: np.ndarray             # [S,4,4] float32, mm, left-model convention (right hands mirrored later)
    hand_idx: np.ndarray          # [S] int64
    frame_of: np.ndarray          # [S] int64 global frame index
    views: np.ndarray             # [S,2] source view per crop
    src_cams: np.ndarray          # [2S,30] float64
    dst_cams: np.ndarray          # [2S,20] float64
    src_offset: np.ndarray        # [2S] int64 byte offset of the view inside the frame buffer
    src_pitch: np.ndarray         # [2S] int32
    intrinsics: np.ndarray        # [2S,3,3] float32
    extrinsics: np.ndarray        # [2S,4,4] float32 (world->crop eye, translation in metres)
    image_bytes: int              # size of the source frame buffer
    rig_cams: np.ndarray = None   # [4,30] float64, the rig's fisheye cameras (pack_fisheye rows)


def make_workload(n_frames: int, frames_per_recording: int, seed: int, hand_model, fk_fn) -> Workload:
    """hand_model: lib.common.hand.HandModel (mm).  fk_fn(joint_angles[N,22], wrist[N,4,4], hand_idx[N])
    -> landmarks [N,21,3] float32 in world mm with the right-hand mirror applied (the GPU FK kernel in
    product code, the oracle in CPU tests)."""
    assert n_frames % frames_per_recording == 0
    rng = np.random.default_rng(seed)
    T, n_rec = frames_per_recording, n_frames // frames_per_recording
    intr, coeffs, rig_c2w = make_rig(rng)
    S = n_rec * 2 * T
    rec = np.repeat(np.arange(n_rec), 2 * T)
    hand = np.tile(np.repeat(np.arange(2), T), n_rec)
    t = np.tile(np.arange(T), n_rec * 2)
    frame_of = rec * T + t
    lim = np.asarray(hand_model.joint_limits, dtype=np.float64)
    ja = rng.uniform(lim[:, 0], lim[:, 1], size=(S, 22)).astype(np.float32)
    ja[:, 20:] = 0
    wrist = np.tile(np.eye(4), (S, 1, 1))
    wrist[:, :3, :3] = random_rotations(rng, S)
    wrist[:, :3, 3] = np.stack([rng.uniform(-120, 120, S), rng.uniform(-90, 90, S), rng.uniform(350, 500, S)], -1)
    wrist = wrist.astype(np.float32)
    # crop points: tracked pose + neutral + open hand (perspective_crop.py:89-115)
    neutral = (lim[:, 0] * 0.5 + lim[:, 1] * 0.5).astype(np.float32)
    poses = np.stack([ja, np.broadcast_to(neutral, ja.shape), np.zeros_like(ja)], 1).reshape(-1, 22)
    lm = fk_fn(poses, np.repeat(wrist, 3, axis=0), np.repeat(hand, 3)).reshape(S, 63, 3)
    # two views per hand: the cameras whose optical axis is closest to the hand (forced 2-view, section 8d)
    centre = lm[:, :21].mean(axis=1).astype(np.float64)
    rel = centre[:, None, :] - rig_c2w[None, :, :3, 3]
    cosang = np.einsum("svj,vj->sv", rel / np.linalg.norm(rel, axis=2, keepdims=True), rig_c2w[:, :3, 2])
    views = np.sort(np.argsort(-cosang, axis=1)[:, :2], axis=1)                 # sort_camera_index=True
    crop_view = views.reshape(-1)
    pts2 = np.repeat(lm, 2, axis=0)
    mirror = np.repeat(hand == 1, 2)
    f, c, crop_c2w, ok = crop_cameras_batched(rig_c2w[crop_view], np.zeros(2 * S), pts2, mirror)
    if not ok.all():
        raise RuntimeError(f"{(~ok).sum()} synthetic crops are degenerate; change the seed")
    rig_cams = np.stack([gpu_warp.pack_fisheye(SRC_W, SRC_H, intr[v, :2], intr[v, 2:], coeffs[v], rig_c2w[v])
                         for v in range(N_VIEWS)])
    src_cams = rig_cams[crop_view]
    dst_cams = np.concatenate([np.stack([f, f, np.full_like(f, c), np.full_like(f, c)], 1),
                               crop_c2w.reshape(-1, 16)], axis=1)
    K = np.zeros((2 * S, 3, 3), np.float32)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 0, 2] = K[:, 1, 2] = c
    K[:, 2, 2] = 1
    E = np.linalg.inv(crop_c2w)
    E[:, :3, 3] *= 0.001                                                        # tracker.py:214-215
    frame_bytes = SRC_H * SRC_W * N_VIEWS                                       # views side by side in one frame
    offs = np.repeat(frame_of, 2).astype(np.int64) * frame_bytes + crop_view.astype(np.int64) * SRC_W
    return Workload(n_rec=n_rec, T=T, joint_angles=ja, wrist=wrist, hand_idx=hand.astype(np.int64),
                    frame_of=frame_of.astype(np.int64), views=views, src_cams=src_cams, dst_cams=dst_cams,
                    src_offset=offs, src_pitch=np.full(2 * S, SRC_W * N_VIEWS, np.int32), intrinsics=K,
                    extrinsics=E.astype(np.float32), image_bytes=n_frames * frame_bytes, rig_cams=rig_cams)
