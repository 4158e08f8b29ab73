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


class TrajectoryShard:
    """
    This rank's output buffer of the path, laid out for the ONE collective: ``buf [2, cap, T, D]`` fp32 with
    ``cap = ceil(B / world)``.  The kernels write positions into ``pos = buf[0, :rows]`` and velocities into
    ``vel = buf[1, :rows]`` (``TrajectoryEngine.trajectory(..., out=(shard.pos, shard.vel))``), so the all-gather sends
    ``buf`` as it lies: no stack before, no copy after.  Ranks that own one episode fewer (ragged split) leave the last
    row of each half unused.
    """

    def __init__(self, num_episodes: int, num_steps: int, num_dof: int, device, rank: Optional[int] = None,
                 world: Optional[int] = None, group=None):
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.num_episodes, self.rank, self.world = int(num_episodes), int(rank), int(world)
        self.cap = -(-self.num_episodes // self.world) if self.num_episodes else 0
        a, b = shard_bounds(self.num_episodes, self.rank, self.world)
        self.start, self.rows = a, b - a
        self.buf = torch.zeros((2, self.cap, int(num_steps), int(num_dof)), dtype=torch.float32, device=device)
        self.pos, self.vel = self.buf[0, : self.rows], self.buf[1, : self.rows]

    def gather(self, out: Optional[torch.Tensor] = None, group=None) -> "GatheredTrajectories":
        """ONE all_gather_into_tensor of ``buf`` (RCCL over xGMI with backend "nccl"); nothing else touches the data"""
        shape = (self.world,) + tuple(self.buf.shape)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.buf.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous fp32 tensor of shape {shape}")
        if self.world == 1 or not dist.is_initialized():
            if out.data_ptr() != self.buf.data_ptr():
                out[0].copy_(self.buf)
        else:
            # the [world * 2, cap, T, D] view is the shape gloo's all_gather_into_tensor insists on; same memory
            dist.all_gather_into_tensor(out.view((self.world * 2,) + tuple(self.buf.shape[1:])), self.buf, group=group)
        return GatheredTrajectories(out, self.num_episodes)


class GatheredTrajectories:
    """
    What every rank holds after the collective: ``buf [world, 2, cap, T, D]``; ``pos`` / ``vel`` are VIEWS
    ``[world, cap, T, D]`` (rank-major: episode ``i`` of rank ``r`` is global episode ``shard_bounds(B, r, world)[0] + i``).
    With an even split ``cap`` rows of every rank are valid; a ragged split leaves ``rows(r) < cap`` for the last ranks.
    ``flat()`` makes the ``[B, T, D]`` copies for a consumer that insists on one contiguous array per quantity.
    """

    def __init__(self, buf: torch.Tensor, num_episodes: int):
        self.buf, self.num_episodes = buf, int(num_episodes)
        self.world, self.cap = int(buf.shape[0]), int(buf.shape[2])
        self.pos, self.vel = buf[:, 0], buf[:, 1]

    def rows(self, rank: int) -> int:
        a, b = shard_bounds(self.num_episodes, rank, self.world)
        return b - a

    def episode(self, i: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(pos [T, D], vel [T, D]) views of global episode ``i``"""
        base, extra = divmod(self.num_episodes, self.world)
        r = i // (base + 1) if i < extra * (base + 1) else extra + (i - extra * (base + 1)) // max(base, 1)
        j = i - shard_bounds(self.num_episodes, r, self.world)[0]
        return self.pos[r, j], self.vel[r, j]

    def __iter__(self):
        """``pos, vel = gather_trajectories(...)`` keeps working (the round-2 return type): the [B, T, D] pair of flat()"""
        return iter(self.flat())

    def flat(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """[B, T, D] copies in global episode order (the only place a copy is made, and only on request)"""
        if self.num_episodes == self.world * self.cap:
            return self.pos.reshape((self.num_episodes,) + tuple(self.pos.shape[2:])), \
                   self.vel.reshape((self.num_episodes,) + tuple(self.vel.shape[2:]))
        idx = [slice(0, self.rows(r)) for r in range(self.world)]
        return torch.cat([self.pos[r, s] for r, s in enumerate(idx)]), torch.cat([self.vel[r, s] for r, s in enumerate(idx)])


def gather_trajectories(pos: torch.Tensor, vel: torch.Tensor, num_episodes: int, group=None) -> GatheredTrajectories:
    """
    The single all-gather of the path (fancy_gym/black_box/black_box_wrapper.py:96-120: episodes are independent, so this
    is the only exchange step).  Zero-copy when ``pos`` / ``vel`` are the two halves of a ``TrajectoryShard`` (the engine
    wrote them there); any other pair of ``[b, T, D]`` tensors is staged into a shard with ONE copy each first.
    Returns views (``GatheredTrajectories``): ``.pos`` / ``.vel`` ``[world, cap, T, D]``.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cap = -(-int(num_episodes) // world) if num_episodes else 0
    item = pos.element_size()
    half = cap * int(pos.shape[1]) * int(pos.shape[2]) * item if pos.dim() == 3 else -1
    adjacent = (pos.dim() == 3 and pos.dtype == torch.float32 and vel.dtype == torch.float32 and pos.shape == vel.shape
                and pos.is_contiguous() and vel.is_contiguous() and vel.data_ptr() == pos.data_ptr() + half
                and pos.untyped_storage().data_ptr() == vel.untyped_storage().data_ptr()
                # ... and the storage really holds the whole [2, cap, T, D] block behind `pos` (a ragged shard of tensors that
                # did not come from a TrajectoryShard may end before it: stage a copy then)
                and pos.storage_offset() * item + 2 * half <= pos.untyped_storage().nbytes())
    sh = TrajectoryShard.__new__(TrajectoryShard)
    sh.num_episodes, sh.rank, sh.world, sh.cap = int(num_episodes), rank, world, cap
    a, b = shard_bounds(num_episodes, rank, world)
    sh.start, sh.rows = a, b - a
    if pos.shape[0] != sh.rows:
        raise ValueError(f"rank {rank} owns {sh.rows} of {num_episodes} episodes, got {pos.shape[0]} rows")
    if adjacent:
        sh.buf = torch.as_strided(pos, (2, cap) + tuple(pos.shape[1:]),
                                  (cap * pos.shape[1] * pos.shape[2], pos.shape[1] * pos.shape[2], pos.shape[2], 1))
    else:
        sh.buf = torch.zeros((2, cap) + tuple(pos.shape[1:]), dtype=torch.float32, device=pos.device)
        sh.buf[0, : sh.rows].copy_(pos)
        sh.buf[1, : sh.rows].copy_(vel)
    sh.pos, sh.vel = sh.buf[0, : sh.rows], sh.buf[1, : sh.rows]
    return sh.gather(group=group)


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
