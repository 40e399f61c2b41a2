#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (build container only).

Run once, here:  python tests/golden/make_goldens.py
Needs /root/reference on disk; the GPU box never runs this.  Only *data* is
written (inputs + the reference's outputs); no reference source is copied.

What can be imported here: lib.models.*, lib.common.{camera,
crop,affine,metric_utils}, lib.batched_dataset.data_transform, load_eval.
What cannot (ModuleNotFoundError: cv2 / pytorch3d / av): lib.tracker.tracker,
lib.common.hand_skinning, lib.tracker.perspective_crop -> no goldens for
cv2.remap and so3_exp_map ("parity unpinned", see oracle/__init__.py).
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

from umetrack_amd import arch  # noqa: E402

torch.set_num_threads(8)
SEED = 20240


def build_reference_model(state):
    from lib.models import feature_extractor as fe, model_loader as ml, skeleton_encoder as se, temporal as tem
    from lib.models.model_opts import ModelOpts
    from lib.models.umetrack_model import UmeTrackModel
    mo = ModelOpts()
    f = fe.FeatureExtractor((96, 96), ModelOpts())
    m = UmeTrackModel(f, tem.create_temporal_model(mo, f.output_feature_sizes),
                      se.SkeletonEncoder([mo.nSkeletonFeatureChannels, *f.output_feature_sizes]),
                      ml._create_regressor(mo, f.output_feature_sizes, True, False),
                      ml._create_regressor(mo, f.output_feature_sizes, False, True))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})   # strict
    return m.eval()


def rand_rigid(rng, t_scale):
    from scipy.spatial.transform import Rotation
    xf = np.eye(4)
    xf[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.6).as_matrix()
    xf[:3, 3] = rng.normal(size=3) * t_scale
    return xf


def make_frame(rng, n):
    img = rng.integers(0, 256, size=(n, 12, 12)).astype(np.float32)
    img = np.kron(img, np.ones((8, 8), np.float32))                       # blocky 96x96
    img = np.clip(img + rng.integers(-20, 21, size=img.shape), 0, 255).astype(np.uint8)
    K = np.zeros((n, 3, 3), np.float32)
    K[:, 0, 0] = K[:, 1, 1] = rng.uniform(120, 420, n)
    K[:, 0, 2] = K[:, 1, 2] = 47.5
    K[:, 2, 2] = 1
    E = np.stack([np.linalg.inv(rand_rigid(rng, 0.25)) for _ in range(n)]).astype(np.float32)
    return img, K, E


def skeleton_m():
    hm = json.load(open(os.path.join(ROOT, "umetrack_amd", "data", "generic_hand_model_small.json")))
    return (np.asarray(hm["joint_rotation_axes"], np.float32),
            np.asarray(hm["joint_rest_positions"], np.float32) * np.float32(0.001))


def model_goldens():
    from lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData
    state = arch.make_synthetic_state(SEED)
    model = build_reference_model(state)
    taps = {}
    model._regressor_k._pose_regression_layers.register_forward_hook(
        lambda m, i, o: taps.__setitem__("raw", o.flatten(1).detach().numpy().copy()))
    model._regressor_u._pose_regression_layers.register_forward_hook(
        lambda m, i, o: taps.__setitem__("raw", o.flatten(1).detach().numpy().copy()))
    orig_backbone = model._feature_extractor._image_backbone
    orig_backbone.register_forward_hook(
        lambda m, i, o: taps.__setitem__("per_view", o.detach().numpy().copy()))
    orig_ffe = model._forward_feature_extractor

    def ffe(fd, sr):
        out = orig_ffe(fd, sr)
        taps["fused"] = out.detach().numpy().copy()
        return out
    model._forward_feature_extractor = ffe
    orig_temp = model._temporal.forward_temporal_features

    def ftf(*a, **k):
        out = orig_temp(*a, **k)
        taps["temporal"] = out.detach().numpy().copy()
        taps["memory"] = model._temporal._mem_features.detach().numpy().copy()
        taps["prev_ext"] = model._temporal._prev_extrinsics.detach().numpy().copy()
        return out
    model._temporal.forward_temporal_features = ftf

    rng = np.random.default_rng(SEED + 1)
    axes, rest = skeleton_m()
    steps = []

    def run(name, n, sample_range, memory_idx, use_memory, hand_idx, skel, head):
        img, K, E = make_frame(rng, n)
        fd = InputFrameData(torch.from_numpy(img.astype(np.float32) / np.float32(255.0)),
                            torch.from_numpy(K), torch.from_numpy(E))
        desc = InputFrameDesc(torch.tensor(sample_range).long(), torch.tensor(memory_idx).long(),
                              torch.tensor(use_memory).bool(), torch.tensor(hand_idx).long())
        rec = {"name": name, "head": head, "images_u8": img, "intrinsics": K, "extrinsics": E,
               "sample_range": np.asarray(sample_range, np.int64),
               "memory_idx": np.asarray(memory_idx, np.int64),
               "use_memory": np.asarray(use_memory, bool), "hand_idx": np.asarray(hand_idx, np.int64)}
        with torch.no_grad():
            if head == "k":
                sa, sr_ = skel
                rec["skel_axes"], rec["skel_rest"] = sa, sr_
                out = model.regress_pose_use_skeleton(
                    fd, desc, InputSkeletonData(torch.from_numpy(sa), torch.from_numpy(sr_)))
            else:
                out = model.regress_pose_pred_skel_scale(fd, desc)
        rec["joint_angles"] = out.joint_angles.numpy()
        rec["wrist_xfs"] = out.wrist_xfs.numpy()
        rec["sigmas"] = out.landmark_uncertainty_sigmas.numpy()
        if out.skel_scales is not None:
            rec["skel_scales"] = out.skel_scales.numpy()
        rec.update({k: v for k, v in taps.items()})
        steps.append(rec)

    one = (axes, rest)
    per3 = (np.stack([axes] * 3), np.stack([rest * s for s in (0.9, 1.0, 1.12)]).astype(np.float32))
    # (iii) a little sequence on ONE model instance: memory off -> on, a hand drops out,
    # memory grows, mixed view counts; (i) all-2-view, (ii) ragged, (iv) broadcast skeleton.
    run("s0_all2v_nomem", 4, [[0, 2], [2, 4]], [0, 1], [False, False], [0, 1], one, "k")
    run("s1_all2v_mem", 4, [[0, 2], [2, 4]], [0, 1], [True, True], [0, 1], one, "k")
    run("s2_one_hand_1v", 1, [[0, 1]], [1], [True], [1], one, "k")
    run("s3_ragged_grow", 5, [[0, 2], [2, 3], [3, 5]], [0, 1, 3], [False, True, False], [0, 1, 0], per3, "k")
    run("s4_scale_head", 6, [[0, 2], [2, 4], [4, 6]], [2, 0, 1], [False, True, True], [1, 0, 1], None, "u")
    run("s5_ragged_mem", 4, [[0, 1], [1, 3], [3, 4]], [3, 2, 0], [True, True, True], [0, 1, 1], per3, "k")

    flat = {"n_steps": np.array(len(steps)), "seed": np.array(SEED)}
    for i, rec in enumerate(steps):
        for k, v in rec.items():
            flat[f"{i}/{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "model_sequence.npz"), **flat)
    spec = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    json.dump(spec, open(os.path.join(HERE, "state_dict_spec.json"), "w"), indent=0)
    print("model goldens:", [(r["name"], float(np.abs(r["wrist_xfs"][:, :3, 3]).max())) for r in steps])


def geometry_goldens():
    import lib.common.camera as rcam
    from lib.common import crop as rcrop
    from lib.batched_dataset import data_transform as rdt
    rng = np.random.default_rng(SEED + 2)
    out = {}
    n_cases = 6
    for i in range(n_cases):
        W, H = (640, 480) if i % 2 == 0 else (512, 512)
        coeffs = list(rng.uniform(-0.02, 0.02, 4)) + list(rng.uniform(-1e-3, 1e-3, 2)) + list(rng.uniform(-0.02, 0.02, 2))
        f = float(rng.uniform(200, 300))
        c2w = rand_rigid(rng, 40.0)
        src = rcam.Fisheye62CameraModel(W, H, (f, f * 1.01), ((W - 1) / 2 + rng.normal(), (H - 1) / 2 + rng.normal()),
                                        coeffs, c2w)
        # points ~ a hand 350-500 mm in front of the camera, off-axis
        centre_eye = np.array([rng.uniform(-150, 150), rng.uniform(-120, 120), rng.uniform(300, 500)])
        if i == 5:
            centre_eye = np.array([420.0, 50.0, 30.0])          # ~86 deg off-axis: part of the crop sees z<0
        pts_eye = centre_eye + rng.normal(size=(63, 3)) * 45
        pts_world = pts_eye @ c2w[:3, :3].T + c2w[:3, 3]
        mirror, angle = bool(i % 2), float([0, 90, -90, 180, 33.0, 0][i])
        dst = rcrop.gen_crop_parameters_from_points(src, pts_world, (96, 96), mirror, camera_angle=angle,
                                                    focal_multiplier=0.95)
        # tracker.py:61-78 executed with the reference's camera classes
        px, py = np.meshgrid(np.arange(dst.width), np.arange(dst.height))
        win = np.column_stack((px.flatten(), py.flatten()))
        eye_src = src.world_to_eye(dst.eye_to_world(dst.window_to_eye(win)))
        w = src.eye_to_window(eye_src)
        w[eye_src[:, 2] < 0] = -1
        w = w.astype(np.float32)
        out.update({f"{i}/src_wh": np.array([W, H]), f"{i}/src_f": np.array(src.f), f"{i}/src_c": np.array(src.c),
                    f"{i}/src_coeffs": np.array(coeffs), f"{i}/src_c2w": c2w, f"{i}/pts_world": pts_world,
                    f"{i}/mirror": np.array(mirror), f"{i}/angle": np.array(angle),
                    f"{i}/dst_f": np.array(dst.f), f"{i}/dst_c": np.array(dst.c),
                    f"{i}/dst_c2w": np.array(dst.camera_to_world_xf),
                    f"{i}/map_x": w[:, 0].reshape(96, 96), f"{i}/map_y": w[:, 1].reshape(96, 96),
                    f"{i}/n_behind": np.array(int((eye_src[:, 2] < 0).sum()))})
    out["n_cases"] = np.array(n_cases)
    np.savez_compressed(os.path.join(HERE, "warp_maps.npz"), **out)
    print("warp goldens: behind-camera pixels per case", [int(out[f"{i}/n_behind"]) for i in range(n_cases)])

    # A1': pinhole -> pinhole float resample
    n, h, w_ = 4, 72, 88
    imgs = rng.integers(0, 256, size=(n, h, w_)).astype(np.float32)
    ext = np.stack([np.linalg.inv(rand_rigid(rng, 0.05)) for _ in range(n)])
    Ks = np.zeros((n, 3, 3))
    Ks[:, 0, 0] = rng.uniform(60, 90, n)
    Ks[:, 1, 1] = Ks[:, 0, 0] * 1.02
    Ks[:, 0, 2], Ks[:, 1, 2], Ks[:, 2, 2] = (w_ - 1) / 2, (h - 1) / 2, 1
    centre = np.array([0.02, -0.01, 0.4])
    xfs, newK, newE = [], [], []
    for v in range(n):
        off = np.array([[0, 0, 0], [0.2, 0, 0], [-0.2, 0.15, 0], [0, -0.18, 0]][v])    # some crops leave the image
        pts = centre + off + rng.normal(size=(21, 3)) * 0.04
        pts_world = (pts - ext[v][:3, 3]) @ ext[v][:3, :3]
        e, k, r = rdt._gen_crop_matrices(ext[v:v + 1], Ks[v:v + 1], pts_world, bool(v % 2), (96, 96))
        xfs.append(r[0]); newK.append(k[0]); newE.append(e[0])
    xfs = np.stack(xfs)
    new = np.zeros((n, 96, 96), np.float32)
    rdt._resample_images_batched(imgs, new, xfs)
    np.savez_compressed(os.path.join(HERE, "resample.npz"), images=imgs, orig_ext=ext, orig_K=Ks, xfs=xfs,
                        new_K=np.stack(newK), new_E=np.stack(newE), out=new)
    print("resample golden: nonzero fraction", float((new != 0).mean()))


def metric_goldens():
    import load_eval as rle
    from lib.common import metric_utils as rmu
    rng = np.random.default_rng(SEED + 3)
    recs = {}
    all_err, all_acc, all_gacc = [], [], []
    for r, T in enumerate((40, 17, 3, 64)):
        gt = rng.normal(size=(2, T, 21, 3)) * 80
        pred = gt + rng.normal(size=gt.shape) * rng.uniform(2, 25)
        valid = rng.uniform(size=(2, T)) > 0.2
        pred[~valid] = 0
        gt[~valid] = 0
        m = rle._compute_metrics(gt, pred, valid)
        recs.update({f"{r}/gt": gt, f"{r}/pred": pred, f"{r}/valid": valid, f"{r}/err": m.keypoint_errors,
                     f"{r}/acc": m.keypoint_accelerations, f"{r}/gt_acc": m.gt_keypoint_accelerations})
        all_err.append(m.keypoint_errors); all_acc.append(m.keypoint_accelerations)
        all_gacc.append(m.gt_keypoint_accelerations)
    e = np.concatenate(all_err)
    pck = rmu.PCK_curve(e, rmu.PCK_THRESHOLDS) * 100.0
    recs.update(n_rec=np.array(4), pck=pck, auc=np.array(float(rmu.normalized_AUC(rmu.PCK_THRESHOLDS, pck))),
                mpjpe=np.array(e.mean()), mpjpa=np.array(np.concatenate(all_acc).mean()),
                gt_mpjpa=np.array(np.concatenate(all_gacc).mean()))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **recs)
    print("metric goldens: mpjpe", float(e.mean()), "auc", float(recs["auc"]))


if __name__ == "__main__":
    model_goldens()
    geometry_goldens()
    metric_goldens()
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))
