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


def test_bench_rccl_report_condenses_the_init_and_tuning_lines(tmp_path):
    """bench.py's self-documentation of the first multi-GPU run: RCCL's own INIT / TUNING log lines (formats of librccl
    2.2x) -> version, channels, transports, and per collective the algorithm / protocol / channel range / bytes / calls"""
    sys.path.insert(0, ROOT)
    import bench
    log = tmp_path / "rccl.log"
    log.write_text(
        "h:1:1 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6\n"
        "h:1:1 [0] NCCL INFO comm:0x1, nRanks:8, nNodes:1, coll channels:28 collnet channels:0, nvls channels:0, p2p channels:32, p2p channels per peer:4\n"
        "h:1:1 [0] NCCL INFO Channel 00/0 : 0[c000] -> 1[d000] via P2P/IPC comm 0x1 nRanks 08\n"
        "h:1:1 [0] NCCL INFO Channel 01/0 : 0[c000] -> 7[f000] via P2P/IPC comm 0x1 nRanks 08\n"
        + "h:1:1 [0] NCCL INFO AllGather: 22937600 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..27}\n" * 3
        + "h:1:1 [0] NCCL INFO AllReduce: 8 Bytes -> Algo TREE proto LL channel{Lo..Hi}={0..0}\n")
    rep = bench.rccl_report(str(log))
    assert rep["version"].startswith("2.26.6") and rep["coll_channels"] == 28 and rep["p2p_channels"] == 32
    assert rep["transports"] == ["P2P/IPC"]
    ag = rep["collectives"][0]
    assert ag == {"collective": "AllGather", "bytes": 22937600, "algo": "RING", "proto": "SIMPLE", "channels": 28, "calls": 3}
    assert rep["collectives"][1]["collective"] == "AllReduce" and rep["collectives"][1]["channels"] == 1
    assert bench.rccl_report(str(tmp_path / "missing.log")) is None and bench.rccl_report(None) is None


def test_bench_reads_the_gpu_state_without_a_child_process(monkeypatch):
    """the clocks / power / temperatures beside the streaming row come from sysfs files read in-process: no subprocess is
    ever spawned (rocm-smi is a `#!/usr/bin/env python3` script: an exec hop this pool forbids under rocprofv3), and a
    machine without the files gets an empty dict, never an error"""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench

    def boom(*a, **k):
        raise AssertionError("bench.gpu_state_sysfs must not spawn a process")
    monkeypatch.setattr(subprocess, "run", boom)
    monkeypatch.setattr(subprocess, "Popen", boom)
    monkeypatch.setattr(subprocess, "call", boom)
    st = bench.gpu_state_sysfs(0)
    assert isinstance(st, dict)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "rocm-smi\"" not in src and "'rocm-smi'" not in src


def test_trajectory_shard_single_process_views_and_episode_lookup():
    """TrajectoryShard / GatheredTrajectories without a process group (world = 1): the gather is the identity, views alias"""
    from fancy_gym_amd.distributed import GatheredTrajectories, TrajectoryShard, gather_trajectories, shard_bounds
    sh = TrajectoryShard(5, 4, 3, "cpu")
    assert sh.rows == 5 and sh.cap == 5 and sh.buf.shape == (2, 5, 4, 3)
    sh.pos.copy_(torch.arange(60, dtype=torch.float32).reshape(5, 4, 3)); sh.vel.copy_(-sh.pos)
    g = sh.gather()
    assert g.pos.shape == (1, 5, 4, 3) and torch.equal(g.pos[0], sh.pos) and torch.equal(g.vel[0], sh.vel)
    p, v = g.flat()
    assert torch.equal(p, sh.pos) and torch.equal(v, sh.vel)
    g2 = gather_trajectories(sh.pos, sh.vel, 5)                       # halves of one shard: zero copy in
    assert torch.equal(g2.buf, g.buf)
    # episode lookup over a ragged 3-rank layout
    buf = torch.zeros((3, 2, 3, 1, 1))
    gg = GatheredTrajectories(buf, 7)
    for i in range(7):
        r = next(k for k in range(3) if shard_bounds(7, k, 3)[0] <= i < shard_bounds(7, k, 3)[1])
        buf[r, 0, i - shard_bounds(7, r, 3)[0], 0, 0] = float(i + 1)
    assert [float(gg.episode(i)[0][0, 0]) for i in range(7)] == [1, 2, 3, 4, 5, 6, 7]
    with pytest.raises(ValueError):
        sh.gather(out=torch.zeros((2, 2, 5, 4, 3)))
