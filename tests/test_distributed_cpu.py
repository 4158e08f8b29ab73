"""
CPU suite, part 3: the N > 1 path (episode sharding + the single all-gather) with world_size 2 on the gloo backend.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from fancy_gym_amd import distributed as D
    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world)
    T, Dof = 6, 3
    full = torch.arange(B * T * Dof, dtype=torch.float32).reshape(B, T, Dof)
    a, b = D.shard_bounds(B, rank, world)
    pos_local = D.shard(full)                       # stands in for the trajectories this rank generated
    assert pos_local.shape[0] == b - a
    vel_local = -pos_local
    # (i) any pair of [b, T, D] tensors: staged into a shard, ONE collective, views back
    g = D.gather_trajectories(pos_local, vel_local, B)
    pos_all, vel_all = g.flat()
    ok = torch.equal(pos_all, full) and torch.equal(vel_all, -full)
    ok = ok and g.pos.shape == (world, -(-B // world), T, Dof) and g.pos.data_ptr() == g.buf.data_ptr()
    ok = ok and all(torch.equal(g.episode(i)[0], full[i]) and torch.equal(g.episode(i)[1], -full[i]) for i in range(B))
    # (ii) the zero-copy route: the "kernels" write into the two halves of a TrajectoryShard, the collective sends the
    # buffer as it lies (the views handed back alias the receive buffer, the send buffer is the shard itself)
    sh = D.TrajectoryShard(B, T, Dof, "cpu")
    ok = ok and sh.rows == b - a and (sh.rows == 0 or sh.pos.data_ptr() == sh.buf.data_ptr())
    sh.pos.copy_(pos_local); sh.vel.copy_(vel_local)
    g2 = D.gather_trajectories(sh.pos, sh.vel, B)
    ok = ok and all(torch.equal(g2.pos[r, :g2.rows(r)], full[slice(*D.shard_bounds(B, r, world))]) for r in range(world))
    ok = ok and all(torch.equal(g2.vel[r, :g2.rows(r)], -full[slice(*D.shard_bounds(B, r, world))]) for r in range(world))
    g3 = sh.gather()
    ok = ok and torch.equal(g3.buf, g2.buf)
    rows = D.all_gather_rows(torch.full((b - a, 2), float(rank)), B)
    ok = ok and rows.shape == (B, 2) and float(rows[:a + 1].min()) >= 0
    expected = torch.cat([torch.full((D.shard_bounds(B, k, world)[1] - D.shard_bounds(B, k, world)[0], 2), float(k))
                          for k in range(world)])
    ok = ok and torch.equal(rows, expected)
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7, 1])
def test_shard_and_all_gather_world2(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_shard_bounds_cover_the_batch_exactly():
    from fancy_gym_amd.distributed import shard_bounds
    for B in (0, 1, 7, 8, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_native_comm_argument_checks_without_a_gpu():
    """include/mpk.h mpk_comm_*: bad arguments are refused with a message before RCCL or a GPU is touched"""
    import ctypes as C
    from fancy_gym_amd import _lib
    lib = _lib.load()
    ident = (C.c_uint8 * _lib.MPK_COMM_ID_BYTES)()
    h = C.c_void_p()
    for rank, world in ((0, 0), (2, 2), (-1, 2)):
        assert lib.mpk_comm_create(ident, rank, world, 0, C.byref(h)) == _lib.MPK_EINVAL
        assert "rank" in _lib.last_error() and not h.value
    assert lib.mpk_comm_create(None, 0, 1, 0, C.byref(h)) == _lib.MPK_EINVAL
    assert lib.mpk_comm_unique_id(None) == _lib.MPK_EINVAL
    assert lib.mpk_allgather(None, None, None, 4, None) == _lib.MPK_EINVAL
    assert lib.mpk_comm_rank(None) == _lib.MPK_EINVAL and lib.mpk_comm_world(None) == _lib.MPK_EINVAL
    lib.mpk_comm_destroy(None)   # no-op


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """bench.py --gpus N under a launcher that created a different number of ranks is an error (exit 2), decided before
    anything touches a GPU -- a silent `n_gpus: 1` line for `--gpus 8` is what round 1 shipped"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()


def test_bench_gpus_n_spawns_a_torch_distributed_run_child(monkeypatch):
    """--gpus N without WORLD_SIZE: the parent builds the torch.distributed.run command line for N ranks on 127.0.0.1 and
    returns the child's exit code (the child is faked here: no GPU in this container)"""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
