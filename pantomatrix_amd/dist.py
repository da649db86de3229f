"""Multi-GPU plumbing for the inference path: clips are independent units, so ranks are replicas —
one process per GPU, clip i handled by rank i % world, NO collective on the data path.  torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) is used only to line ranks up and to reduce timings.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env)."""
    rank, _local, world = env_world()
    if world == 1 and "RANK" not in os.environ:      # plain `python bench.py`: no process group at all
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def shard_clips(n_clips: int, rank: int, world: int):
    """Indices of the clips rank `rank` owns: round-robin, i -> rank i % world."""
    return list(range(rank, n_clips, world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(units_local: float, elapsed_local: float, device="cpu") -> float:
    """Whole-job rate: units processed by ALL ranks / the slowest rank's time."""
    return sum_over_ranks(units_local, device) / max_over_ranks(elapsed_local, device)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
