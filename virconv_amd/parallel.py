"""One-process-per-GPU data parallelism: frames are independent units, sharded across ranks with no
data-path collective; the only exchange is the DDP gradient all-reduce (RCCL over xGMI through torch's 'nccl' backend;
'gloo' on CPU for tests).  Mirrors tools/train.py:63-65,141 + pcdet/utils/common_utils.py:141-154.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run). -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_frames(frame_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Round-robin frame sharding (what DistributedSampler does without shuffling, datasets/__init__.py:66-71)."""
    return [f for k, f in enumerate(frame_ids) if k % world == rank]


def wrap_ddp(model: torch.nn.Module, device=None) -> torch.nn.Module:
    """DDP wrap (train.py:141).  One bucket holds the whole backbone (1.7 MB): a single fused all-reduce per step."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    kw = dict(bucket_cap_mb=64, broadcast_buffers=False)
    if device is not None and torch.device(device).type == "cuda":
        idx = torch.device(device).index
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[idx], **kw)
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
