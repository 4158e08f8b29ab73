"""
Multi-GPU layer of the path: episodes are independent units, so a batch shards contiguously across ranks with NO
data-path collective; the only exchange step is the optional collection of the generated trajectories on every rank
(one all-gather over RCCL / xGMI; backend "nccl" IS RCCL on ROCm, "gloo" for the CPU tests).
One process per GPU; rendezvous via the usual RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT environment.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """initialise torch.distributed from the environment (no-op for a single process); returns (rank, world, local)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_bounds(num_episodes: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous split: rank r owns rows [start, stop); the first (B % world) ranks get one extra episode"""
    base, extra = divmod(int(num_episodes), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(x: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """this rank's rows of a [B, ...] tensor"""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    a, b = shard_bounds(x.shape[0], rank, world)
    return x[a:b]


def all_gather_rows(local: torch.Tensor, num_episodes: int, group=None) -> torch.Tensor:
    """
    Collect the row shards of every rank: [b_r, ...] on each rank -> [B, ...] on every rank, in rank order.
    Ragged shards are padded to the largest shard so that ONE all_gather_into_tensor moves everything.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    cap = -(-int(num_episodes) // world)
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != cap:
        send = torch.zeros((cap,) + tail, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * cap,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if num_episodes == world * cap:
        return recv
    parts = []
    for r in range(world):
        a, b = shard_bounds(num_episodes, r, world)
        parts.append(recv[r * cap: r * cap + (b - a)])
    return torch.cat(parts, dim=0)


def gather_trajectories(pos: torch.Tensor, vel: torch.Tensor, num_episodes: int, group=None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """the single all-gather of the path: (pos | vel) stacked so that one collective carries both"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return pos, vel
    both = all_gather_rows(torch.stack([pos, vel], dim=1), num_episodes, group)   # [B, 2, T, D]
    return both[:, 0].contiguous(), both[:, 1].contiguous()
