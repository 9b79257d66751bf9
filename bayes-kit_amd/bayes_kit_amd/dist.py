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


# Collectives issued through this module and diagnostics.py since import, by kind: lets a caller state how
# many a summary cost (bench.py: `collectives_per_summary`) from calls that really happened.
collective_calls = {"all_gather": 0, "all_reduce": 0, "all_to_all": 0}


# A process group of ONE rank keeps its summaries local (nothing to exchange).  force_collectives (or the environment variable
# BK_DIST_FORCE_COLLECTIVES=1) sends them through the group all the same: every collective the N > 1 path issues -- all_gather
# with a tensor list, all_to_all_single with count lists, all_reduce -- then runs on RCCL with a single-rank group on ONE GPU,
# which is how a one-GPU box exercises the backend the 8-GPU run will use (tests/test_gpu_multirank.py).
force_collectives = bool(os.environ.get("BK_DIST_FORCE_COLLECTIVES"))


def collectives_active(group=None) -> bool:
    """Whether summaries go through the process group: more than one rank, or a group whose use is forced."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or force_collectives


def host_staged(t: torch.Tensor, group=None) -> torch.Tensor:
    """`t` as the process group can move it.  RCCL moves device memory; gloo implements all_gather /
    all_to_all for host memory only, so a device tensor is staged through the host there (the CPU tests,
    and bench.py's ranks-share-one-GPU mode)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu()
    return t


def all_gather(t: torch.Tensor, group=None):
    """Every rank's `t` (same shape on all ranks), as a list in rank order, on t's device."""
    world = dist.get_world_size(group)
    send = host_staged(t.contiguous(), group)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    collective_calls["all_gather"] += 1
    return parts if send.device == t.device else [p.to(t.device) for p in parts]


def all_to_all_single(recv: torch.Tensor, send: torch.Tensor, recv_counts=None, send_counts=None, group=None):
    """dist.all_to_all_single into `recv` (rows split by the count lists, or evenly)."""
    s_ = host_staged(send.contiguous(), group)
    if s_.device == send.device:
        dist.all_to_all_single(recv, s_, recv_counts, send_counts, group=group)
    else:
        r_ = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r_, s_, recv_counts, send_counts, group=group)
        recv.copy_(r_)
    collective_calls["all_to_all"] += 1
    return recv


def gather_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of `t` over the ranks of `group`, accumulated in rank order (deterministic: the result does
    not depend on the collective's reduction order).  ONE all_gather.  No-op without a process group."""
    if not collectives_active(group):
        return t
    parts = all_gather(t, group)
    out = parts[0].clone()
    for p in parts[1:]:
        out += p
    return out


def sum_over_ranks(value: float, device=None, group=None) -> float:
    """Scalar sum over ranks (e.g. total ESS, accepted counts)."""
    if not collectives_active(group):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    collective_calls["all_reduce"] += 1
    return float(t.item())
