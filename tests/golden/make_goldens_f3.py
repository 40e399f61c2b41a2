#!/usr/bin/env python3
"""Goldens for the torch_data path; build container only.

  python tests/golden/make_goldens_f3.py

1. writes a tiny synthetic dataset tests/golden/torch_data/testing/{mono,labels}.torch.{idx,bin}
   plus a ragged float32 pair with this repo's writer (the reference ships no writer and no data);
2. reads it back with the REFERENCE's TorchIdx (lib/data_utils/idxbinfile.py) and stores what it
   saw in torch_idx.npz - this pins the file layout;
3. runs the REFERENCE's preprocess (lib/batched_dataset/data_transform.py:388-397) on every sample
   and stores its outputs in torch_data_preprocess.npz - this pins crop cameras, resampling and
   the hand-model handling.
Only data is written; no reference source is copied.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import fk as ofk  # noqa: E402
from umetrack_amd.lib.data_utils.idxbinfile import write_torch_bin  # noqa: E402

T, V, H, W = 3, 2, 120, 160
N_SAMPLES = 4


def look_at(eye, target):
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross([0.0, 1.0, 0.0], z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
    return c2w


def make_sample(rng, hm_json, hand):
    hm = {k: np.asarray(v) for k, v in hm_json.items() if isinstance(v, list)}
    lim = hm["joint_limits"]
    centre = np.array([rng.uniform(-40, 40), rng.uniform(-30, 30), rng.uniform(380, 460)])
    eyes = [np.array([-30.0, 0.0, 0.0]), np.array([35.0, 5.0, 0.0])]
    ja = rng.uniform(lim[:, 0], lim[:, 1], size=(T, 22)).astype(np.float32)
    ja[:, 20:] = 0
    wrist = np.tile(np.eye(4), (T, 1, 1))
    from scipy.spatial.transform import Rotation
    for t in range(T):
        wrist[t, :3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.5).as_matrix()
        wrist[t, :3, 3] = centre + rng.normal(size=3) * 8
    lm = ofk.skin_landmarks(hm, ja, wrist)                                     # (T,21,3) mm, left-hand convention
    if hand == 1:
        lm = lm * np.array([-1.0, 1.0, 1.0])                                   # a right hand seen in the world
    ext = np.stack([np.stack([np.linalg.inv(look_at(e, centre + rng.normal(size=3) * 25)) for e in eyes])
                    for _ in range(T)])
    K = np.zeros((T, V, 3, 3))
    K[..., 0, 0] = K[..., 1, 1] = rng.uniform(150, 190, size=(T, V))
    K[..., 0, 2], K[..., 1, 2], K[..., 2, 2] = (W - 1) / 2, (H - 1) / 2, 1
    img = rng.integers(0, 256, size=(T, V, H // 8, W // 8)).astype(np.float32)
    img = np.kron(img, np.ones((8, 8), np.float32))
    img = np.clip(img + rng.integers(-12, 13, size=img.shape), 0, 255).astype(np.uint8)
    gen = dict(hm_json)
    scaled = dict(hm_json)
    s = float(rng.uniform(0.9, 1.1))
    scaled["joint_rest_positions"] = (np.asarray(hm_json["joint_rest_positions"]) * s).tolist()
    scaled["landmark_rest_positions"] = (np.asarray(hm_json["landmark_rest_positions"]) * s).tolist()
    scaled["hand_scale"] = s
    labels = {
        "extrinsics": ext.tolist(), "intrinsics": K.tolist(), "enclosing_points": lm.tolist(),
        "hand": [float(hand)] * T, "hand_model": scaled, "wrist": wrist.tolist(), "joint_angles": ja.tolist(),
        "solved_wrist_xfs": (wrist + rng.normal(size=wrist.shape) * 1e-3).tolist(),
        "solved_joint_angles": (ja + rng.normal(size=ja.shape).astype(np.float32) * 1e-2).tolist(),
        "generic_hand_model": gen, "pinch": rng.uniform(0, 1, size=(T, 4)).tolist(),
    }
    return img, labels


def main():
    rng = np.random.default_rng(31337)
    hm_json = json.load(open(os.path.join(ROOT, "umetrack_amd", "data", "generic_hand_model_small.json")))
    folder = os.path.join(HERE, "torch_data", "testing")
    os.makedirs(folder, exist_ok=True)
    samples = [make_sample(rng, hm_json, hand=i % 2) for i in range(N_SAMPLES)]
    write_torch_bin(os.path.join(folder, "mono.torch.idx"), [s[0] for s in samples], "uint8")
    write_torch_bin(os.path.join(folder, "labels.torch.idx"), [s[1] for s in samples], "object")
    ragged = [rng.normal(size=shape).astype(np.float32) for shape in ((2, 3), (4,), (1, 2, 2))]
    write_torch_bin(os.path.join(HERE, "torch_data", "ragged.torch.idx"), ragged, "float32")

    # ---- the reference reads what we wrote ----
    from lib.data_utils.idxbinfile import TorchIdx as RefIdx
    out = {}
    for name, path in (("mono", os.path.join(folder, "mono.torch.idx")), ("labels", os.path.join(folder, "labels.torch.idx")),
                       ("ragged", os.path.join(HERE, "torch_data", "ragged.torch.idx"))):
        idx = RefIdx(path)
        out[f"{name}_len"] = len(idx)
        out[f"{name}_uniform"] = idx.is_uniform
        out[f"{name}_shape"] = np.array(idx.shape if idx.shape is not None else [], dtype=np.int64)
        out[f"{name}_dtype"] = str(idx.dtype)
        out[f"{name}_offsets"] = np.array([idx.byte_offset(i) for i in range(len(idx))] + [idx.byte_offset(-1)], dtype=np.int64)
        if idx.is_uniform:
            data = idx.read_bin()
        else:
            # the reference's lazy non-uniform view asserts on a plain range (dataset_util.py:155);
            # its per-frame accessor is what its datasets actually call (async_dataset.py:146)
            raw = open(idx.bin_path, "rb").read()
            data = [idx.view_buffer_at(i, raw) for i in range(len(idx))]
        if name == "mono":
            assert np.array_equal(data, np.stack([s[0] for s in samples]))
            out["mono_checksum"] = np.array([int(np.asarray(data[i], dtype=np.int64).sum()) for i in range(len(idx))])
        elif name == "ragged":
            for i in range(len(idx)):
                assert np.array_equal(data[i], ragged[i])
                out[f"ragged_{i}"] = np.array(data[i])
        else:
            for i in range(len(idx)):
                assert data[i]["hand"] == samples[i][1]["hand"]
            out["labels_keys"] = np.array(sorted(data[0].keys()))
    np.savez_compressed(os.path.join(HERE, "torch_idx.npz"), **out)

    # ---- the reference pre-processes every sample ----
    from lib.batched_dataset.data_transform import preprocess as ref_preprocess
    from lib.data_utils import bundles as ref_bundles
    mono_idx, lab_idx = RefIdx(os.path.join(folder, "mono.torch.idx")), RefIdx(os.path.join(folder, "labels.torch.idx"))
    mono = mono_idx.read_bin()
    lab_raw = open(lab_idx.bin_path, "rb").read()
    labs = [lab_idx.view_buffer_at(i, lab_raw) for i in range(len(lab_idx))]
    pre = {}
    items = []
    for i in range(N_SAMPLES):
        mi, mt = ref_preprocess({"mono": np.array(mono[i]), "labels": labs[i]}, (96, 96))
        items.append((mi, mt))
        pre[f"s{i}_left_images"] = mi.left_images.numpy()
        pre[f"s{i}_intrinsics"] = mi.intrinsics.numpy()
        pre[f"s{i}_extrinsics_xf"] = mi.extrinsics_xf.numpy()
        pre[f"s{i}_hand_idx"] = mi.hand_idx.numpy()
        pre[f"s{i}_joint_angles"] = mi.orig_pose_data.joint_angles.numpy()
        pre[f"s{i}_wrist_xfs"] = mi.orig_pose_data.wrist_xfs.numpy()
        pre[f"s{i}_axes"] = mi.orig_pose_data.left_hand_model.joint_rotation_axes.numpy()
        pre[f"s{i}_rest"] = mi.orig_pose_data.left_hand_model.joint_rest_positions.numpy()
        pre[f"s{i}_landmarks_rest"] = mi.orig_pose_data.left_hand_model.landmark_rest_positions.numpy()
        pre[f"s{i}_solved_rest"] = mi.s_solved_pose_data.left_hand_model.joint_rest_positions.numpy()
        pre[f"s{i}_target_wrist"] = mt.gt_skel_targets.wrist_xfs.numpy()
        pre[f"s{i}_target_scale"] = np.asarray(mt.gt_skel_targets.skel_scales.numpy())
        pre[f"s{i}_pinch"] = mt.gt_skel_targets.pinch_prediction.numpy()
    batch_in, _ = ref_bundles.collate(items)
    pre["batch_left_images_shape"] = np.array(batch_in.left_images.shape)
    pre["batch_axes_shape"] = np.array(batch_in.orig_pose_data.left_hand_model.joint_rotation_axes.shape)
    pre["n_samples"] = N_SAMPLES
    np.savez_compressed(os.path.join(HERE, "torch_data_preprocess.npz"), **pre)
    cov = [float((pre[f's{i}_left_images'] > 0).mean()) for i in range(N_SAMPLES)]
    print("wrote goldens; crop coverage per sample:", cov)


if __name__ == "__main__":
    main()
