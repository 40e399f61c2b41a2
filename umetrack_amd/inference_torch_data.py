"""torch_data inference (BASELINE.json configs[0]): sequences from
``{field}.torch.idx/.bin`` -> GPU pre-processing -> per-step model calls with the temporal memory
-> keypoint error in mm.  Mirrors run_inference_torch_data.py:39-207; the async loader and
DataLoader workers of the reference are replaced by a plain loop over memory-mapped samples,
because the per-pixel work they parallelised now runs on the GPU.

  python -m umetrack_amd.inference_torch_data --data UmeTrack_data/torch_data --weights pretrained_weights.torch
"""
import argparse
import os
from typing import List, Optional, Tuple

import torch

from .lib.batched_dataset.data_transform import ModelInput, ModelTarget, preprocess
from .lib.common.hand import mirrored_hand_model
from .lib.common.hand_skinning import skin_landmarks
from .lib.data_utils import bundles
from .lib.data_utils.split import Split
from .lib.data_utils.torch_dataset import find_dataset
from .lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData


def unpack_batched_data(model_input: ModelInput, seq_mode: str
                        ) -> List[Tuple[InputFrameData, InputFrameDesc, InputSkeletonData]]:
    """Batch of sequences (bs, T, views, ...) -> one model call per time step
    (run_inference_torch_data.py:39-85): memory row = batch index, memory used from step 1 on."""
    if seq_mode not in ("multiv", "singlev"):
        raise ValueError(f"Unknown sequence mode: {seq_mode}")
    nv = 2 if seq_mode == "multiv" else 1
    images = model_input.left_images
    bs, seq_len = images.shape[0], images.shape[1]
    dev = images.device
    hm = model_input.orig_pose_data.left_hand_model
    sample_range = (torch.arange(bs, device=dev)[:, None] * nv + torch.tensor([0, nv], device=dev)).long()
    steps = []
    for t in range(seq_len):
        frame = InputFrameData(left_images=images[:, t, 0:nv].flatten(0, 1),
                               intrinsics=model_input.intrinsics[:, t, 0:nv].flatten(0, 1),
                               extrinsics_xf=model_input.extrinsics_xf[:, t, 0:nv].flatten(0, 1))
        desc = InputFrameDesc(sample_range=sample_range, memory_idx=torch.arange(bs, device=dev),
                              use_memory=torch.full((bs,), t > 0, dtype=torch.bool, device=dev),
                              hand_idx=model_input.hand_idx[:, t].long())
        skel = InputSkeletonData(joint_rotation_axes=hm.joint_rotation_axes[:, t],
                                 joint_rest_positions=hm.joint_rest_positions[:, t])
        steps.append((frame, desc, skel))
    return steps


def eval_batch(model, model_input: ModelInput, model_target: ModelTarget, cur_mode: str, use_skel: bool,
               device) -> torch.Tensor:
    """Mean keypoint error in mm per sequence (run_inference_torch_data.py:88-135); FK of the
    prediction and of the ground truth run in the GPU kernel."""
    hand_model = mirrored_hand_model(model_input.orig_pose_data.left_hand_model, model_input.hand_idx == 1)
    outputs = []
    for frame, desc, skel in unpack_batched_data(model_input, cur_mode):
        frame, desc, skel = bundles.to_device((frame, desc, skel), device)
        if use_skel:
            outputs.append(model.regress_pose_use_skeleton(frame, desc, skel))
        else:
            assert cur_mode == "multiv", "Skeleton scale prediction requires multiv data"
            outputs.append(model.regress_pose_pred_skel_scale(frame, desc))
    joint_angles = torch.stack([o.joint_angles for o in outputs], 1)          # (bs, T, 22)
    wrist_xfs = torch.stack([o.wrist_xfs for o in outputs], 1)
    target = model_target.gt_skel_targets
    gt = skin_landmarks(hand_model, target.joint_angles, target.wrist_xfs)
    pred = skin_landmarks(hand_model, joint_angles, wrist_xfs)
    return (gt - pred).norm(dim=-1).mean(dim=(1, 2)) * 1000


def run(data_roots, model, device, batch_size: int = 160, crop_size=(96, 96), splits=(Split.TEST,),
        seq_mode: str = "multiv", use_skel: bool = True, max_batches: Optional[int] = None):
    """{split: mean keypoint error in mm}; incomplete last batches are dropped like the
    reference's Sampler(drop_last=True)."""
    datasets = find_dataset(data_roots, ["mono", "labels"])
    result = {}
    for split, dataset in datasets.items():
        if split not in splits or len(dataset) < batch_size:
            continue
        errors = []
        n_batches = len(dataset) // batch_size
        if max_batches is not None:
            n_batches = min(n_batches, max_batches)
        for b in range(n_batches):
            items = [preprocess(dataset[b * batch_size + i], crop_size, device) for i in range(batch_size)]
            model_input, model_target = bundles.collate(items)
            model.reset_memory()
            errors.append(eval_batch(model, model_input, model_target, seq_mode, use_skel, device))
        result[split] = float(torch.cat(errors).mean())
        print(f"Keypoint errors ({split.value}): {result[split]} mm")
    return result


def main():
    from .lib.models.model_loader import load_pretrained_model
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", nargs="+", required=True, help="torch_data roots (real / synthetic)")
    ap.add_argument("--weights", required=True)
    ap.add_argument("--batch-size", type=int, default=160)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("umetrack_amd has no CPU path: an MI355X is required")
    device = torch.device("cuda", 0)
    model = load_pretrained_model(args.weights).eval().to(device)
    for k, v in run(args.data, model, device, args.batch_size).items():
        print(k, v)


if __name__ == "__main__":
    main()
