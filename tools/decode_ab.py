#!/usr/bin/env python3
"""Head outputs (fusion/temporal transforms, decode) of two builds of the library, bitwise, on random regressor rows and a
recurrent sequence; decode launch time.   python3 tools/decode_ab.py tools/bin/libumt_dec_old.so   (GPU box; experiments)"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umetrack_amd import _native as nat, arch  # noqa: E402
from umetrack_amd.lib.common.hand import load_generic_hand_model, scaled_hand_model  # noqa: E402
from umetrack_amd.lib.models.model_loader import model_from_state_dict  # noqa: E402
from umetrack_amd.lib.models.umetrack_model import InputFrameData, InputFrameDesc, InputSkeletonData  # noqa: E402


def run(path):
    from scipy.spatial.transform import Rotation
    dev = torch.device("cuda", 0)
    m = model_from_state_dict(arch.make_synthetic_state(5)).to(dev)
    rng = np.random.default_rng(1)
    outs = {}
    hm = scaled_hand_model(load_generic_hand_model(), 0.001)
    skel = InputSkeletonData(hm.joint_rotation_axes.to(dev), hm.joint_rest_positions.to(dev))
    for bs in (2, 4096):
        n = 2 * bs
        E = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
        E[:, :3, :3] = Rotation.from_rotvec(rng.normal(size=(n, 3)) * 0.5).as_matrix()
        E[:, :3, 3] = rng.normal(size=(n, 3)) * 0.2
        K = torch.zeros(n, 3, 3)
        K[:, 0, 0] = K[:, 1, 1] = torch.from_numpy(rng.uniform(150, 400, n).astype(np.float32))
        K[:, 0, 2] = K[:, 1, 2] = 47.5
        K[:, 2, 2] = 1
        img = torch.from_numpy(rng.random((n, 96, 96), dtype=np.float32))
        fd = InputFrameData(img.to(dev), K.to(dev), torch.from_numpy(E).to(dev))
        for step in range(3):
            desc = InputFrameDesc(torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(bs),
                                  torch.full((bs,), step > 0), torch.arange(bs) % 2)
            for head in ("k", "u"):
                o = m.regress_pose_use_skeleton(fd, desc, skel) if head == "k" else m.regress_pose_pred_skel_scale(fd, desc)
                outs[f"{bs}.{step}.{head}.ja"] = o.joint_angles.cpu().numpy()
                outs[f"{bs}.{step}.{head}.wx"] = o.wrist_xfs.cpu().numpy()
                outs[f"{bs}.{step}.{head}.sg"] = o.landmark_uncertainty_sigmas.cpu().numpy()
        mem, prev = m.memory_state()
        outs[f"{bs}.mem"] = mem.cpu().numpy()
        m.reset_memory()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        desc = InputFrameDesc(torch.arange(bs)[:, None] * 2 + torch.tensor([0, 2]), torch.arange(bs), torch.zeros(bs, dtype=torch.bool),
                              torch.arange(bs) % 2)
        for _ in range(3):
            m.regress_pose_use_skeleton(fd, desc, skel)
        e0.record()
        for _ in range(10):
            m.regress_pose_use_skeleton(fd, desc, skel)
        e1.record()
        torch.cuda.synchronize()
        print(f"{os.environ.get('UMT_LIB_PATH', 'base'):42s} bs {bs}: forward {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us", flush=True)
    np.savez(path, **outs)


if len(sys.argv) > 2 and sys.argv[1] == "child":
    run(sys.argv[2])
    sys.exit(0)
subprocess.run([sys.executable, __file__, "child", "/tmp/dec_other.npz"], check=True,
               env=dict(os.environ, UMT_LIB_PATH=os.path.join(ROOT, sys.argv[1])), timeout=300)
run("/tmp/dec_base.npz")
a, b = np.load("/tmp/dec_base.npz"), np.load("/tmp/dec_other.npz")
bad = [k for k in a.files if a[k].tobytes() != b[k].tobytes()]
print(f"{len(a.files)} arrays compared;", "all bit-identical" if not bad else f"DIFFERENT: {bad}")
