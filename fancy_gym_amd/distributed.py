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


class NativeComm:
    """
    The same all-gather without torch.distributed in the data path: libmpk's RCCL communicator (include/mpk.h
    ``mpk_comm_*`` / ``mpk_allgather``).  Rank 0 draws the unique id; the 128 id bytes travel through
    ``exchange(id_bytes_or_None) -> id_bytes`` -- by default a torch.distributed object broadcast (host side, any
    backend), but any host-side channel works (a file, MPI, an environment variable for world == 1).
    """

    def __init__(self, rank: int, world: int, device: int, exchange=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        buf = (C.c_uint8 * _lib.MPK_COMM_ID_BYTES)()
        if self.rank == 0:
            _lib.check(self._lib.mpk_comm_unique_id(buf))
        ident = bytes(buf)
        if self.world > 1:
            if exchange is None:
                box = [ident if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ident = box[0]
            else:
                ident = exchange(ident if self.rank == 0 else None)
        h = C.c_void_p()
        raw = (C.c_uint8 * _lib.MPK_COMM_ID_BYTES).from_buffer_copy(ident)
        _lib.check(self._lib.mpk_comm_create(raw, self.rank, self.world, self.device, C.byref(h)))
        self._h = h

    def all_gather(self, shard: torch.Tensor, out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """[...] fp32 on this rank -> [world, ...] on every rank, ONE ncclAllGather on ``stream`` (default: current)"""
        from . import _lib
        if shard.dtype != torch.float32 or not shard.is_cuda or not shard.is_contiguous():
            raise ValueError("all_gather needs a contiguous fp32 device tensor")
        if out is None:
            out = torch.empty((self.world,) + tuple(shard.shape), dtype=torch.float32, device=shard.device)
        elif out.numel() != self.world * shard.numel() or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 tensor of world * shard.numel() elements")
        s = torch.cuda.current_stream(shard.device) if stream is None else stream
        _lib.check(self._lib.mpk_allgather(self._h, shard.data_ptr(), out.data_ptr(), shard.numel(), s.cuda_stream))
        return out

    def gather_trajectories(self, pos_vel: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``pos_vel`` [2, b, T, D] (the kernels write pos into [0], vel into [1]) -> [world, 2, b, T, D]"""
        return self.all_gather(pos_vel, out)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.mpk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
