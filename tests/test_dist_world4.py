"""run_eval(..., world=4) + aggregate_metrics(reduce_over_ranks=True) under gloo with four CPU ranks:
whole recordings of uneven length per rank, one all-reduce of the 108-word partial vector, every rank ends with
the global numbers.  The tracker and the device metric kernel are stubbed (no GPU here): what is under test is
the sharding, the dataset walk, the label/frame reader, the result files and the reduction - the parts a real
8-GPU run adds to the single-GPU path (the reference's fan-out: run_eval_known_skeleton.py:113-115)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, zlib
sys.path.insert(0, sys.argv[1])
data_root, out_root = sys.argv[2], sys.argv[3]
import numpy as np
import torch
from oracle import metrics as omx
from umetrack_amd import dist as udist, eval as ev
from umetrack_amd.lib.common import camera
from umetrack_amd.lib.common.hand import load_generic_hand_model
from umetrack_amd.lib.tracker import video_pose_data as vpd

rank, world, _ = udist.init_from_env("gloo")
LENGTHS = [37, 5, 64, 12, 23, 9, 50]                   # frames per recording: uneven on purpose
H, W = 6, 8

def kp_from_labels(lab, seed):
    """Deterministic stand-ins for landmarks: a function of the labels the reader delivered."""
    T = len(lab)
    gt = lab.wrist_transforms[:, :, None, :3, 3] + np.arange(21)[None, None, :, None] * np.array([1.0, 2.0, 3.0])
    gt = np.transpose(gt, (1, 0, 2, 3)).copy()                      # [2,T,21,3]
    rng = np.random.default_rng(seed)
    tracked = gt + rng.normal(size=gt.shape) * 7.0
    valid = np.transpose(lab.hand_confidences > 0, (1, 0)).copy()
    return tracked, gt, valid

if rank == 0:                                                       # one writer, then everybody reads
    hm = load_generic_hand_model()
    cams = [camera.PinholePlaneCameraModel(W, H, (10.0, 10.0), (3.5, 2.5), []) for _ in range(4)]
    for i, T in enumerate(LENGTHS):
        rng = np.random.default_rng(100 + i)
        d = os.path.join(data_root, f"user{i % 3:02d}", "testing" if i != 3 else "training")
        os.makedirs(d, exist_ok=True)
        wr = np.tile(np.eye(4), (T, 2, 1, 1)); wr[..., :3, 3] = rng.normal(size=(T, 2, 3)) * 30
        conf = (rng.uniform(size=(T, 2)) > 0.25).astype(np.float64)
        vpd.write_recording(os.path.join(d, f"rec{i}.mono"), rng.integers(0, 256, (T, H, 4 * W), dtype=np.uint8), cams,
                            [0.0] * 4, hm, rng.normal(size=(T, 2, 22)), wr, conf)
udist.barrier()

seen_frames = []
def fake_track_sequence(stream, model, output_path=None, override=False):
    n = 0
    for frame, gt_tracking in stream:                               # the real reader: frames split into 4 views
        assert len(frame.views) == 4 and frame.views[0].image.shape == (H, W)
        n += 1
    assert n == len(stream)
    seen_frames.append(n)
    tracked, gt, valid = kp_from_labels(stream.labels, zlib.crc32(os.path.basename(output_path).encode()))
    ev._save_result(output_path, tracked, gt, valid)
    return np.linalg.norm((gt - tracked)[valid], axis=-1).mean(axis=-1)

def cpu_partials(gt, pred, valid):                                  # stands in for csrc/metrics.hip
    a = omx.aggregate([(gt.numpy(), pred.numpy(), valid.numpy())])
    return torch.tensor([a["sum_err"], a["n_err"], a["sum_acc"], a["n_acc"], a["sum_gt_acc"], a["n_valid"], a["n_total"]]
                        + a["pck_counts"].tolist(), dtype=torch.float64)

ev.track_sequence = fake_track_sequence
ev.mu.metric_partials = cpu_partials
errs, files = ev.run_eval(data_root, out_root, model=None, protocol="known", rank=rank, world=world)
m = ev.aggregate_metrics(out_root, files, reduce_over_ranks=True, verbose=False)

# what every rank must have ended with: the numbers over ALL testing recordings (index 3 is a training folder)
test_ids = [i for i in range(len(LENGTHS)) if i != 3]
recs = []
for i in test_ids:
    lab = vpd._load_hand_pose_labels(os.path.join(data_root, f"user{i % 3:02d}", "testing", f"rec{i}.json"))
    tracked, gt, valid = kp_from_labels(lab, zlib.crc32(f"rec{i}.npy".encode()))
    recs.append((gt, tracked, valid))
full = omx.aggregate(recs)
assert m.n_total == full["n_total"] == 2 * sum(LENGTHS[i] for i in test_ids), (m.n_total, full["n_total"])
assert m.n_valid == full["n_valid"]
for key in ("mpjpe", "mpjpa", "gt_mpjpa", "auc"):
    assert abs(getattr(m, key) - full[key]) < 1e-9 * max(1.0, abs(full[key])), (key, getattr(m, key), full[key])
# sharding: whole recordings, disjoint, complete, balanced by size (largest-first greedy)
mine = sorted(int(os.path.basename(f)[3:-4]) for f in files)
gathered = [None] * world
torch.distributed.all_gather_object(gathered, (mine, int(sum(seen_frames))))
allrec = sorted(i for g, _ in gathered for i in g)
assert allrec == test_ids, gathered
loads = [n for _, n in gathered]
assert max(loads) <= 64 and min(loads) >= 23, loads              # 64 | 50 | 37 | 23+12+9+5 would be optimal by frames
assert errs.shape[0] == sum(int(r[2].sum()) for r, i in zip(recs, test_ids) if i in mine)
sys.stdout.write(f"EVAL OK rank {rank} of {world}: recordings {mine}\n")
udist.barrier()
'''


def test_run_eval_and_metric_reduction_world_size_4_gloo(tmp_path):
    script = tmp_path / "worker4.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", str(script), ROOT,
                          str(tmp_path / "data"), str(tmp_path / "results")],
                         capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for r in range(4):
        assert f"EVAL OK rank {r} of 4" in out.stdout
