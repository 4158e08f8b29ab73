#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MP hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic episodes: ONE launch of the fused kernel
(basis contraction on MFMA + ProDMP boundary conditions + PD tracking-controller actions) for BASELINE config 2:
ProDMP, 7 DoF, 5 basis (+goal) per DoF, 100 steps, batch 4096 per GPU, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--eager] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Multi-GPU: episodes shard across ranks (independent units, no data-path collective):
"scaling": "weak", value = all ranks' trajectories / max-over-ranks time.  The optional all-gather of the generated
trajectories over RCCL/xGMI (north_star) is timed separately and reported under "allgather".

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent -- before it makes a
single GPU call -- runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>` as a
CHILD process and exits with its code (never an exec of a process that touched the GPU).  A WORLD_SIZE that disagrees
with `--gpus` is an error (exit 2), not a warning.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# BASELINE cfg2 (fancy_gym/envs/registry.py:105-128 + envs/mujoco/box_pushing/mp_wrapper.py:12-28)
CFG = dict(num_dof=7, num_basis=5, dt=0.02, duration=2.0, tau=1.5, alpha_phase=3.0, basis_bandwidth_factor=2.0,
           basis_alpha=10.0)
P_GAINS = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
D_GAINS = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])
T_STEPS, D, P = 100, 7, 42
# algorithmic bytes per trajectory: 224 B in + 2*T*D*4 B (pos, vel) + T*D*4 B (torques)
BYTES_PER_TRAJ = 224 + 2 * T_STEPS * D * 4 + T_STEPS * D * 4
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096, help="episodes per GPU per step")
    ap.add_argument("--eager", action="store_true", help="one host launch per step instead of one hipGraph of K steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="(accepted for older scripts: the two-stream leg is off by default since round 6)")
    ap.add_argument("--overlap", action="store_true",
                    help="also time the K launches on two streams (consecutive steps may overlap).  Off by default: measured SLOWER than the "
                         "plain line in round 5 (4.29 vs 4.44 x 10^8 trajectories/s) -- two launches of one kernel that each fill the chip "
                         "share its write path, they do not hide each other's ramp")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-streaming", action="store_true", help="skip the HBM-streaming roofline row (B = 262144)")
    ap.add_argument("--streaming-batch", type=int, default=262144)
    return ap.parse_args()


def pmc_traffic(batch: int):
    """
    HBM bytes per launch from the PMC counters of the committed profile (profiles/pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md); null when no
    profile exists for this batch size.  The live run cannot read PMCs itself.
    """
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tab = json.load(f)
        row = tab.get(str(batch))
        return float(row["hbm_bytes_per_launch"]) if row else None
    except Exception:
        return None


def cpu_baseline(seconds: float, batch: int):
    """The numpy oracle ("port") timed on this box's host cores, one thread, bounded sample of the same workload."""
    from oracle import mp_oracle as O
    pc = O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0)
    bc = O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10)
    tc = O.TrajCfg("prodmp", action_dim=7)
    tabs = O.prodmp_tables(pc, bc, np.float32)
    rng = np.random.default_rng(0)
    params = rng.standard_normal((batch, P)).astype(np.float32)
    ip = rng.uniform(-1, 1, (batch, D)).astype(np.float32)
    iv = np.zeros((batch, D), np.float32)

    def one(bsl):
        pos, vel = O.get_trajectory(pc, bc, tc, params[bsl], 2.0, 0.02, 0.0, ip[bsl], iv[bsl], tables=tabs)
        O.rollout(pos, vel, "motor", P_GAINS, D_GAINS, -1.0, 1.0, "static", 0.02, ip[bsl].astype(np.float64),
                  iv[bsl].astype(np.float64))

    # (i) batched: the whole batch per call (best-effort CPU)
    one(slice(0, batch))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds * 0.6:
        one(slice(0, batch)); n += batch
    batched = n / (time.perf_counter() - t0)
    # (ii) reference-style: B = 1 per call in a Python loop (black_box_wrapper.py:96-120 execution model)
    m, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < seconds * 0.4:
        i = m % batch
        one(slice(i, i + 1)); m += 1
    ref_style = m / (time.perf_counter() - t1)
    # (iii) BASELINE.md section 3 "CPU-batched": the same restated path with torch-CPU ops on every host core
    # A batch of 4096 is small for a 256-thread box (oversubscribed it runs slower than one core), so the thread count is
    # scanned and the best one reported next to nproc.
    from oracle.mp_torch_cpu import ProDMPBatchedCPU
    ncores = os.cpu_count() or 1
    fn = ProDMPBatchedCPU(pc, bc, tc, 2.0, 0.02)
    tp, tip, tiv = torch.from_numpy(params), torch.from_numpy(ip), torch.from_numpy(iv)
    pg, dg = torch.from_numpy(P_GAINS), torch.from_numpy(D_GAINS)
    cpos, cvel = tip.double(), tiv.double()
    best = None
    scan = {}
    for nt in sorted({ncores, max(1, ncores // 2), max(1, ncores // 4), 32, 16, 8, 4, 1} & set(range(1, ncores + 1))):
        torch.set_num_threads(nt)
        for _ in range(3):
            fn(tp, tip, tiv, 0.0, pg, dg, -1.0, 1.0, cpos, cvel)
        reps = []
        t2 = time.perf_counter()
        while len(reps) < 20 or (time.perf_counter() - t2 < seconds * 0.04 and len(reps) < 2000):
            r0 = time.perf_counter()
            fn(tp, tip, tiv, 0.0, pg, dg, -1.0, 1.0, cpos, cvel)
            reps.append(time.perf_counter() - r0)
            if time.perf_counter() - t2 > seconds * 0.08:
                break
        rate = batch / float(np.median(reps))
        scan[nt] = rate
        if best is None or rate > best[1]:
            best = (nt, rate, len(reps))
    torch.set_num_threads(1)
    all_cores = best[1]
    return {"value": batched, "unit": "trajectories/s", "cores": 1, "kind": "port",
            "sample": f"numpy oracle (fp32), ProDMP 7-DoF/5 basis/100 steps + PD actions: {n} trajectories in "
                      f"batches of {batch} over {seconds * 0.6:.0f} s; reference-style B=1 Python loop: "
                      f"{ref_style:.0f} trajectories/s over {m} calls",
            "reference_style_b1": {"value": ref_style, "unit": "trajectories/s", "cores": 1, "calls": m},
            "all_cores": {"value": all_cores, "unit": "trajectories/s", "cores": best[0], "nproc": ncores, "kind": "port",
                          "sample": f"oracle/mp_torch_cpu.py (torch-CPU einsum), batch of {batch} per call, median of "
                                    f"{best[2]} calls at the best of the scanned thread counts",
                          "threads_scan": {str(k): v for k, v in scan.items()}}}


def rccl_debug_setup(rank: int):
    """
    Make the first multi-GPU run document itself: unless the caller chose an NCCL_DEBUG level, RCCL logs its INIT and
    TUNING lines (rings / channels at communicator creation; algorithm, protocol and channel range per collective) into a
    per-process file, which `rccl_report` condenses into the "allgather" object.  Nothing is logged inside the headline's
    timed region (it makes no RCCL call).  Returns the log path or None.
    """
    if "NCCL_DEBUG" in os.environ and os.environ["NCCL_DEBUG"].upper() not in ("VERSION", "WARN"):
        return os.environ.get("NCCL_DEBUG_FILE")          # the caller is debugging: leave everything alone
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"mpk_bench_rccl.{os.getpid()}.log")
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
    os.environ["NCCL_DEBUG_FILE"] = path
    return path


def rccl_report(path):
    """algorithm / protocol / channels RCCL chose, parsed from its own log (formats of librccl 2.2x: `<Coll>: <n> Bytes ->
    Algo <a> proto <p> channel{Lo..Hi}={l..h}`, `comm:..., coll channels:<n> ... p2p channels:<n>`, `Channel xx/yy : a[..] -> b[..] via <transport>`)"""
    import re
    if not path or not os.path.exists(path):
        return None
    try:
        with open(path, errors="replace") as f:
            text = f.read()
    except OSError:
        return None
    rep = {}
    m = re.search(r"coll channels:\s*(\d+).*?p2p channels:\s*(\d+)", text)
    if m:
        rep["coll_channels"], rep["p2p_channels"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"RCCL version\s*:?\s*([^\s]+)", text) or re.search(r"NCCL version\s*([^\s]+)", text)
    if m:
        rep["version"] = m.group(1)
    transports = sorted(set(re.findall(r"Channel \d+(?:/\d+)? : \d+\[[0-9a-f]+\] -> \d+\[[0-9a-f]+\] (?:\[\w+\] )?via (\S+)", text)))
    if transports:
        rep["transports"] = transports
    colls = {}
    for name, nbytes, algo, proto, lo, hi in re.findall(
            r"(\w+): (\d+) Bytes -> Algo (\S+) proto (\S+) channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\}", text):
        key = f"{name}/{nbytes}B"
        c = colls.setdefault(key, {"collective": name, "bytes": int(nbytes), "algo": algo, "proto": proto,
                                   "channels": int(hi) - int(lo) + 1, "calls": 0})
        c["calls"] += 1
    if colls:
        rep["collectives"] = sorted(colls.values(), key=lambda c: -c["bytes"])[:6]
    rep["log_lines"] = text.count("\n")
    return rep or None


def gpu_state_sysfs(dev_index: int = 0):
    """
    Clocks / power / temperatures of the GPU this process runs on, read from sysfs files IN-PROCESS (no child process:
    `rocm-smi` is a `#!/usr/bin/env python3` script, and spawning it from a process running under rocprofv3 would be the
    exec-after-GPU-init hop this pool forbids).  Best effort: missing files are skipped, never an error.
    """
    import glob
    out = {}
    try:
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            pass
        cards = []
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
            if "-" in os.path.basename(c):
                continue
            d = os.path.join(c, "device")
            try:
                with open(os.path.join(d, "vendor")) as f:
                    if f.read().strip() != "0x1002":
                        continue
            except OSError:
                continue
            cards.append(d)
        if not cards:
            return out
        dev = next((d for d in cards if want and os.path.basename(os.path.realpath(d)) == want), cards[min(dev_index, len(cards) - 1)])
        out["sysfs_device"] = os.path.basename(os.path.realpath(dev))

        def rd(path):
            try:
                with open(path) as f:
                    return f.read().strip()
            except OSError:
                return None
        for name in ("sclk", "mclk", "fclk", "socclk"):
            txt = rd(os.path.join(dev, "pp_dpm_" + name))
            if txt:
                cur = [ln for ln in txt.splitlines() if ln.rstrip().endswith("*")] or txt.splitlines()[-1:]
                digits = "".join(ch for ch in cur[0].split(":")[-1] if ch.isdigit())
                if digits:
                    out[name + "_mhz"] = int(digits)
        for key in ("gpu_busy_percent", "mem_busy_percent"):
            v = rd(os.path.join(dev, key))
            if v and v.lstrip("-").isdigit():
                out[key] = int(v)
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for key in ("power1_average", "power1_input"):
                v = rd(os.path.join(hw, key))
                if v and v.isdigit():
                    out["package_power_w"] = int(v) / 1e6
                    break
            for tin in glob.glob(os.path.join(hw, "temp*_input")):
                v = rd(tin)
                label = rd(tin.replace("_input", "_label")) or os.path.basename(tin).replace("_input", "")
                if v and v.lstrip("-").isdigit():
                    out["temp_" + label.lower().replace(" ", "_") + "_c"] = int(v) / 1000.0
    except Exception:  # noqa: BLE001 - the sample is optional
        pass
    return out


def synth_inputs(B: int, seed: int):
    """synthetic MP parameters of the named shape: params ~ N(0,1), init_pos ~ U(-1,1), init_vel = 0;
    seed 0 on one GPU, 1000 + rank when sharded"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    params = torch.randn((B, P), generator=g, dtype=torch.float32)
    init_pos = torch.rand((B, D), generator=g, dtype=torch.float32) * 2 - 1
    init_vel = torch.zeros((B, D), dtype=torch.float32)
    return params, init_pos, init_vel


def spawn_ranks(n: int) -> int:
    """start the n ranks of `--gpus n` as a child torch.distributed.run; the parent never touches the GPU"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        print("[bench] --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] error: --gpus {args.gpus} but WORLD_SIZE {world}: launch one rank per GPU "
                  f"(--nproc-per-node {args.gpus}) or drop WORLD_SIZE and let bench.py start them", file=sys.stderr)
        sys.exit(2)
    # MPK_BENCH_FORCE_DIST=1 runs the collective code path (RCCL) even with a single rank, to exercise it on a 1-GPU box
    force_dist = os.environ.get("MPK_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        # RCCL prints a version banner on stdout at NCCL_DEBUG=VERSION; stdout must carry the JSON line only.  Its INIT /
        # TUNING lines go to a per-process file instead (rccl_debug_setup) and come back condensed in "allgather.rccl".
        rccl_log = rccl_debug_setup(rank)
        import torch.distributed as dist
        # MPK_BENCH_BACKEND=gloo + more ranks than GPUs: a rehearsal of the N > 1 code path on a 1-GPU box (ranks share
        # the device; RCCL refuses that, gloo does not care).  The driver's runs use the default: one rank per GPU, RCCL.
        backend = os.environ.get("MPK_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if force_dist and "MASTER_ADDR" not in os.environ:
            # a port nobody holds right now (two forced one-rank runs on one box must not meet on a fixed number)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        rccl_log = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    from fancy_gym_amd import RolloutSpec, TrajectoryEngine
    from fancy_gym_amd import _lib
    import ctypes as C

    eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=local_rank, **CFG)
    assert eng.num_steps == T_STEPS and eng.num_params == P
    B, K, W = args.batch, args.steps, args.warmup
    params, init_pos, init_vel = (t.to(dev) for t in synth_inputs(B, 1000 + rank if world > 1 else 0))
    c_pos, c_vel = init_pos.double().contiguous(), init_vel.double().contiguous()
    pos, vel, act = (torch.empty((B, T_STEPS, D), dtype=torch.float32, device=dev) for _ in range(3))
    spec = RolloutSpec("motor", D, P_GAINS, D_GAINS, -1.0, 1.0, plant="static")

    lib = _lib.load()
    h, rcfg = eng._h, C.byref(spec.c)
    ptrs = [t.data_ptr() for t in (params, init_pos, init_vel)]
    outs = [t.data_ptr() for t in (pos, vel, act)]
    cp, cv = c_pos.data_ptr(), c_vel.data_ptr()

    def step(stream_ptr):
        rc = lib.mpk_trajectory_actions(h, ptrs[0], ptrs[1], ptrs[2], 0.0, rcfg, cp, cv, outs[0], outs[1], outs[2], B,
                                        stream_ptr)
        if rc != 0:
            raise RuntimeError(_lib.last_error())

    def barrier():
        if dist is not None:
            dist.barrier()

    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    for _ in range(W):
        step(sp)
    torch.cuda.synchronize()

    # ---- the timed region: EXACTLY K steps ------------------------------------------------------------------------
    launch = "eager"
    graph = None
    untimed_replays = 0
    if not args.eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(stream)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(K):
                        step(side.cuda_stream)
            stream.wait_stream(side)
            torch.cuda.synchronize()
            # untimed: the first replay uploads the executable graph; keep replaying until the GPU has been busy for
            # ~30 ms so that short runs (small K) are measured at the same clocks as long ones
            t_busy = time.perf_counter()
            untimed_replays = 0
            while untimed_replays < 1 or (time.perf_counter() - t_busy < 0.03 and untimed_replays < 1000):
                graph.replay()
                torch.cuda.synchronize()
                untimed_replays += 1
            launch = "hipgraph"
        except Exception as e:  # pragma: no cover - depends on the runtime
            print(f"[bench] hipGraph capture failed ({e}); falling back to eager launches", file=sys.stderr)
            graph = None
    # ---- K steps, wall clock between barrier + synchronize on both sides ----------------------------------------------
    def k_steps():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(K):
                step(sp)

    # every rank: barrier -> synchronize -> t0 -> K steps -> synchronize -> t1.  The closing barrier sits OUTSIDE the
    # clock: the MAX over ranks below already accounts for the slowest rank, and an RCCL barrier (a 1-element all-reduce
    # + host sync, tens of us) inside a 0.2 ms region would tax the N-rank line with something the 1-rank line never pays.
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_steps()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    # the kernel's average launch duration for the roofline line: the same K steps once more, bracketed by HIP events on
    # the launch stream (kept out of the wall-clock region above: two event records cost a short run several percent)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    k_steps()
    ev1.record(stream)
    torch.cuda.synchronize()
    kern_avg = ev0.elapsed_time(ev1) * 1e-3 / K
    per_rank = None
    if dist is not None:
        # outside the timed region, no new collective inside it: every rank's own wall-clock sample and its event-timed kernel
        # average, so that a reader of the N-rank line can tell host jitter (one rank's wall sample off, its kernel time not)
        # from imbalance (a rank's kernel time off) without a re-run
        mine = torch.tensor([elapsed, kern_avg], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = {"per_rank_ms": [float(x[0].item()) * 1e3 for x in every],
                    "per_rank_kernel_avg_us": [float(x[1].item()) * 1e6 for x in every]}
        per_rank["kernel_avg_us_max"] = max(per_rank["per_rank_kernel_avg_us"])
        per_rank["event_timed_value"] = world * B / (per_rank["kernel_avg_us_max"] * 1e-6)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * K / elapsed

    # ---- optional: generation + all-gather of (pos | vel) over RCCL/xGMI ------------------------------------------
    # Failures are made COLLECTIVE: everything that can fail locally (allocation, the first launch) happens before the
    # first collective of the leg and is agreed on with a MIN all-reduce, so a rank never leaves its peers blocked in a
    # collective; a failure after the collectives have started ends the rank non-zero (the launcher then ends the job).
    allgather = None
    if dist is not None and not args.no_allgather:
        from fancy_gym_amd.distributed import TrajectoryShard
        Kg = max(10, min(K, 200))
        local_err = None
        try:
            shard = TrajectoryShard(world * B, T_STEPS, D, dev, rank=rank, world=world)   # the kernel writes (pos | vel) here
            full = torch.empty((world,) + tuple(shard.buf.shape), dtype=torch.float32, device=dev)
            sp0, sp1 = shard.pos.data_ptr(), shard.vel.data_ptr()

            def step_into_shard():
                rc = lib.mpk_trajectory_actions(h, ptrs[0], ptrs[1], ptrs[2], 0.0, rcfg, cp, cv, sp0, sp1, outs[2], B, sp)
                if rc != 0:
                    raise RuntimeError(_lib.last_error())
            step_into_shard()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 - agreed on below
            local_err = str(e)[:200]
            print(f"[bench] all-gather leg: rank {rank} cannot take part: {local_err}", file=sys.stderr)
        okf = torch.tensor([0 if local_err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0:
            allgather = {"error": local_err or "another rank could not allocate / launch; leg skipped on every rank"}
        else:
            try:
                for _ in range(5):
                    step_into_shard(); shard.gather(out=full)
                barrier(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(Kg):
                    step_into_shard(); shard.gather(out=full)
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t1
                barrier()
                t = torch.tensor([e2], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e2 = float(t.item())
                # what arrived: an exact, order-independent checksum (sum of the fp32 bit patterns as int64) of every
                # rank's own shard, exchanged on the host, against the same checksum of the slice rank 0 received
                def bits_sum(t):
                    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
                mine = torch.tensor([bits_sum(shard.buf)], dtype=torch.int64, device=dev)
                sums = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(sums, mine)
                shard_sums = [int(x.item()) for x in sums]
                gathered_ok = all(bits_sum(full[r]) == shard_sums[r] for r in range(world))
                allgather = {"value": world * B * Kg / e2, "unit": "trajectories/s", "steps": Kg,
                             "ms_per_step": e2 / Kg * 1e3,
                             "bytes_gathered_per_gpu_per_step": int((world - 1) * shard.buf.numel() * 4),
                             "via": f"torch.distributed all_gather_into_tensor ({'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()}), "
                                    "zero-copy: the kernels write into the send buffer (distributed.TrajectoryShard)",
                             "gathered_equals_shards": bool(gathered_ok), "shard_checksums": shard_sums}
                if os.environ.get("MPK_BENCH_NATIVE_COMM") == "1":
                    # opt-in: the same leg through libmpk's own RCCL communicator (mpk_comm_* / mpk_allgather, include/mpk.h)
                    from fancy_gym_amd.distributed import NativeComm
                    comm = NativeComm(rank, world, local_rank)
                    for _ in range(5):
                        step_into_shard(); comm.all_gather(shard.buf, out=full, stream=stream)
                    barrier(); torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    for _ in range(Kg):
                        step_into_shard(); comm.all_gather(shard.buf, out=full, stream=stream)
                    torch.cuda.synchronize()
                    e3 = time.perf_counter() - t2
                    barrier()
                    t = torch.tensor([e3], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    e3 = float(t.item())
                    allgather["native"] = {"value": world * B * Kg / e3, "ms_per_step": e3 / Kg * 1e3, "via": "mpk_allgather"}
                    comm.close()
            except Exception as e:  # noqa: BLE001
                # collectives of this leg have started: peers may be waiting in one.  Leave non-zero so that the launcher
                # (torch.distributed.run) ends every rank instead of letting them wait for the collective's timeout.
                print(f"[bench] all-gather leg failed on rank {rank} after its collectives started: {e}", file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)
            rccl = rccl_report(rccl_log)
            if rccl is not None and rank == 0:
                allgather["rccl"] = rccl

    headline_kernel = eng.last_kernel()
    # ---- the same K steps with consecutive launches allowed to OVERLAP: two capture streams, alternating output sets ----
    # Every step is still a full B-episode launch writing all three arrays (to set i % 2); step i + 1 no longer waits for
    # step i's last store, so its dispatch ramp and input loads (time-to-first-store, ~1.8 us of the 8.2: profiles/
    # r02_headline_trace.md) run under step i's drain.  Reported BESIDE the serial number: `value` stays the serial one.
    two_stream = None
    if graph is not None and args.overlap and not args.no_overlap and world == 1:    # a one-GPU diagnostic: N > 1 lines carry no extra collectives for it
        try:
            outs_b_t = [torch.empty((B, T_STEPS, D), dtype=torch.float32, device=dev) for _ in range(3)]   # kept alive
            outs_b = [t_.data_ptr() for t_ in outs_b_t]
            sets = (outs, outs_b)
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            sa.wait_stream(stream)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.stream(sa):
                with torch.cuda.graph(g2, stream=sa):
                    sb.wait_stream(sa)                                  # fork
                    for i in range(K):
                        st_, o = (sa, sets[0]) if i % 2 == 0 else (sb, sets[1])
                        rc = lib.mpk_trajectory_actions(h, ptrs[0], ptrs[1], ptrs[2], 0.0, rcfg, cp, cv, o[0], o[1], o[2], B,
                                                        st_.cuda_stream)
                        if rc != 0:
                            raise RuntimeError(_lib.last_error())
                    sa.wait_stream(sb)                                  # join
            stream.wait_stream(sa)
            torch.cuda.synchronize()
            for _ in range(max(2, min(untimed_replays, 50))):
                g2.replay()
            torch.cuda.synchronize()
            reps = []
            for _ in range(5):
                barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                g2.replay()
                torch.cuda.synchronize()
                reps.append(time.perf_counter() - t0)
            e_ov = sorted(reps)[len(reps) // 2]
            if dist is not None:
                t = torch.tensor([e_ov], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e_ov = float(t.item())
            two_stream = {"value": world * B * K / e_ov, "unit": "trajectories/s", "ms_per_step": e_ov / K * 1e3,
                          "launch": "hipgraph of the same K launches on two streams (step i on stream i % 2, output set i % 2): "
                                    "consecutive steps may overlap; every launch writes all three arrays of B episodes",
                          "timing": "wall clock of one replay, median of 5"}
            del g2, outs_b_t
        except Exception as e:  # noqa: BLE001 - an extra, never the headline
            print(f"[bench] two-stream leg failed: {e}", file=sys.stderr)
    # ---- the same kernel family where the outputs really stream to HBM (B = 262144: 2.2 GB written per launch) ---------
    streaming = None
    if rank == 0 and world == 1 and not args.no_streaming:
        try:
            Bs = args.streaming_batch
            gs = torch.Generator(device="cpu").manual_seed(1)
            sp_ = torch.randn((Bs, P), generator=gs, dtype=torch.float32).to(dev)
            sip = (torch.rand((Bs, D), generator=gs, dtype=torch.float32) * 2 - 1).to(dev)
            siv = torch.zeros((Bs, D), dtype=torch.float32, device=dev)
            scp, scv = sip.double().contiguous(), siv.double().contiguous()
            so = [torch.empty((Bs, T_STEPS, D), dtype=torch.float32, device=dev) for _ in range(3)]

            def sstep():
                rc_ = lib.mpk_trajectory_actions(h, sp_.data_ptr(), sip.data_ptr(), siv.data_ptr(), 0.0, rcfg,
                                                 scp.data_ptr(), scv.data_ptr(), so[0].data_ptr(), so[1].data_ptr(),
                                                 so[2].data_ptr(), Bs, sp)
                if rc_ != 0:
                    raise RuntimeError(_lib.last_error())
            # the shader clock needs ~20 ms of streaming load to settle under the package power cap (tools/clock_probe.py:
            # 449 us for the first 40 launches, 422 after), so the row is warmed for ~0.3 s; one sysfs sample taken with
            # launches in flight records the clocks / power / temperatures the number was measured at (boxes differ)
            smi = {}
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.3:
                for _ in range(10):
                    sstep()
                if not smi and time.perf_counter() - t_w > 0.15:
                    smi = gpu_state_sysfs(local_rank)
                torch.cuda.synchronize()
            Ks = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(Ks):
                sstep()
            e1.record(stream)
            torch.cuda.synchronize()
            ks_avg = e0.elapsed_time(e1) * 1e-3 / Ks
            ach = BYTES_PER_TRAJ * Bs / ks_avg / 1e9
            # the box's own ceiling beside it: a plain fill of the same three output arrays (boxes differ by up to 30 %
            # on HBM-streaming launches with identical clocks; a slow box shows here too, a slow kernel does not)
            for o in so:
                o.fill_(1.0)
            e0.record(stream)
            for _ in range(10):
                for o in so:
                    o.fill_(1.0)
            e1.record(stream)
            torch.cuda.synchronize()
            fill_gbs = 10 * sum(o.numel() * 4 for o in so) / (e0.elapsed_time(e1) * 1e-3) / 1e9
            streaming = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(Bs), "kernel": eng.last_kernel(),
                         "kernel_avg_us": ks_avg * 1e6, "batch": Bs, "launches": Ks,
                         "algorithmic_bytes_per_launch": BYTES_PER_TRAJ * Bs,
                         "trajectories_per_s": Bs / ks_avg, "box_fill_GBps": fill_gbs,
                         "gpu_state_during_warmup": smi or None}
            del sp_, sip, siv, scp, scv, so
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] streaming row failed: {e}", file=sys.stderr)

    line = None
    if rank == 0:
        achieved = BYTES_PER_TRAJ * B / kern_avg / 1e9
        out = {
            "metric": "MP trajectories/sec (7-DoF, 5 basis, 100 steps) at 1/2/4/8 GPUs; % roofline",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: ProDMP 7-DoF, 5 basis(+goal), 100 steps, exp phase tau=1.5, alpha=10; "
                                   "trajectory (pos, vel) + PD tracking-controller actions, fused",
                       "batch_per_gpu": B, "global_batch": world * B, "launch": launch, "untimed_graph_replays": untimed_replays,
                       "sharding": f"dp{world} (independent episodes, no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(B),
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc passes of this command on an earlier "
                                           "call (tools/pmc_traffic.sh), committed; not re-measured in this run",
                         "kernel": headline_kernel, "kernel_avg_us": kern_avg * 1e6,
                         "algorithmic_bytes_per_launch": BYTES_PER_TRAJ * B},
        }
        if per_rank is not None:
            # N > 1 only: `value` = world * B * K / max(per_rank_ms); event_timed_value = world * B / kernel_avg_us_max
            out.update(per_rank)
        if two_stream is not None:
            out["two_stream"] = two_stream
        if streaming is not None:
            out["roofline_streaming"] = streaming
        if allgather is not None:
            out["allgather"] = allgather
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, B)
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    if dist is not None:
        dist.destroy_process_group()
        if rccl_log and os.path.basename(rccl_log).startswith("mpk_bench_rccl."):
            try:
                os.remove(rccl_log)
            except OSError:
                pass
    # the JSON line is the LAST thing on stdout: native libraries (RCCL) buffer their own stdout writes until exit
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
