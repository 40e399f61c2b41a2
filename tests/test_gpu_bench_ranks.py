"""bench.py with more than one rank, rehearsed on one GPU: two processes under torch.distributed.run with the gloo
backend, both on cuda:0 (--same-device).  RCCL itself needs a multi-GPU node; what this holds is everything around it:
the launch contract (RANK / WORLD_SIZE from the environment), frames sharded per rank, the barrier + max-over-ranks
timing, the all-reduce of the metric partials inside the timed step, the collective's self-description and rank 0's
single JSON line.  (A device-resident partial vector under a CPU backend once stalled this very path.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--same-device", "--frames", "256", "--chunk-frames", "256", "--cpu-hands", "0", "--no-pcie",
           "--no-x3", "--per-frame-frames", "4"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["hand_frames_per_gpu"] == 512
    col = line["collective"]
    assert col["backend"] == "gloo" and col["world"] == 2 and col["allreduce_us"] > 0
    assert 0 < col["ms_per_step_min_rank"] <= col["ms_per_step_max_rank"]
    # value counts the hand-frames of BOTH ranks over the slowest rank's time
    assert abs(line["value"] - 2 * 512 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["frac"] > 0 and line["mpjpe_mm"] > 0
