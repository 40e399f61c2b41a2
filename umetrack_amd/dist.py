"""One process per GPU; the only collective on the hot path is the metric reduction.

Frames / recordings are independent (run_eval_known_skeleton.py:113-115 fans out per
recording), so ranks never exchange activations.  Each rank reduces its landmarks to the
108-word partial vector of csrc/metrics.hip and one all-reduce(sum) (RCCL over xGMI on
GPUs, gloo in CPU tests) combines them - under 1 KB, latency-bound.
"""
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"     # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)       # one communicator per GPU, bound early
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_recordings(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Whole recordings per rank (temporal state and MPJPA need a recording on one GPU):
    longest-first greedy bin packing by frame count."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    bins: List[List[int]] = [[] for _ in range(world)]
    load = [0] * world
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += lengths[i]
    return [sorted(b) for b in bins]


def all_reduce_partials(partials: torch.Tensor) -> torch.Tensor:
    """Sum the float64[108] metric partials over all ranks (in place); identity for one process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if partials.is_cuda and dist.get_backend() != "nccl":
            # a CPU backend (gloo rehearsals on a GPU host): reduce a host copy; gloo's own handling of device tensors
            # stalled when two ranks shared one GPU
            host = partials.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            partials.copy_(host)
        else:
            dist.all_reduce(partials, op=dist.ReduceOp.SUM)
    return partials


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_max_over_ranks(value: float, device=None) -> Tuple[float, float]:
    """(min, max) of a per-rank scalar, so that a multi-GPU line shows load imbalance, not only the slowest rank."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value, value
    t = torch.tensor([value, -value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(-t[1].item()), float(t[0].item())


def describe(device=None, reps: int = 20) -> dict:
    """What the collective layer of this run actually is: backend, rank count and the measured latency of the one
    collective on the path (all-reduce of the 108-word float64 partial vector), mean of `reps` after a warm-up.
    Every rank must call it (it runs collectives); single-process runs report world 1 and no timing."""
    import time
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return {"backend": None, "world": 1, "allreduce_us": None}
    t = torch.zeros(108, dtype=torch.float64, device=device if device is not None else "cpu")
    on_gpu = t.is_cuda

    def sync():
        if on_gpu:
            torch.cuda.synchronize(t.device)
    dist.all_reduce(t)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(t)
    sync()
    us = (time.perf_counter() - t0) / reps * 1e6
    lo, hi = min_max_over_ranks(us, device)
    return {"backend": dist.get_backend(), "world": dist.get_world_size(), "allreduce_us": hi, "allreduce_us_min_rank": lo,
            "allreduce_words": 108, "allreduce_reps": reps}
