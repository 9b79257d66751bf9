"""Chain sharding across the GPUs of a node (one process per GPU).

Chains are independent, so HMC / MALA / DRGHMC need NO data-path collective: rank r simply
owns a contiguous block of global chain ids and seeds chain g with Philox key (seed, g),
which makes every chain's trajectory identical whatever the number of ranks.  The only
collective is the tiny all_gather inside the R-hat / ESS summaries (diagnostics.py), RCCL
over xGMI on the GPU box, gloo in the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard(total_chains: int, rank: int = None, world: int = None):
    """(first global chain id, number of chains) of this rank: contiguous blocks, the
    remainder spread over the first ranks."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(int(total_chains), world)
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def init_from_env(backend: str = None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (as set by
    torch.distributed.run); returns (rank, local_rank, world).  backend defaults to nccl
    (= RCCL on ROCm) when a GPU is visible, else gloo."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def sum_over_ranks(value: float, device=None, group=None) -> float:
    """Scalar sum over ranks (e.g. total ESS, accepted counts)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return float(t.item())
