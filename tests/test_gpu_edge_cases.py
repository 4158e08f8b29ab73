"""
GPU suite, part 3: edge cases -- every DoF count / basis count / horizon shape class of the tile machinery (ragged last
episode group, partial last row tile, unaligned outputs -> generic store path, D or K beyond the MFMA kernel's limits ->
per-episode kernel), empty batches, bound clipping, range errors, and both work decompositions.
"""
import dataclasses

import numpy as np
import pytest
import torch

from fancy_gym_amd import RolloutSpec, TrajectoryEngine, _lib
from oracle import mp_oracle as O
from tests.test_gpu_trajectory import CFG2, close, fd_atol, inputs, make_engine

pytestmark = pytest.mark.gpu


def cfg_for(mp, D, nb, T, dt=0.02, **tkw):
    dur = T * dt
    if mp == "prodmp":
        return (O.PhaseCfg("exp", tau=dur * 0.75, alpha_phase=3.0), O.BasisCfg("prodmp", num_basis=nb, alpha=10),
                O.TrajCfg("prodmp", action_dim=D, **tkw), dt, dur)
    if mp == "dmp":
        return (O.PhaseCfg("exp", tau=dur, alpha_phase=2.0), O.BasisCfg("rbf", num_basis=nb),
                O.TrajCfg("dmp", action_dim=D, alpha=25.0, weights_scale=0.8, goal_scale=1.2), dt, dur)
    return (O.PhaseCfg("linear", tau=dur), O.BasisCfg("zero_rbf", num_basis=nb, num_basis_zero_start=1,
                                                      num_basis_zero_goal=1),
            O.TrajCfg("promp", action_dim=D, weights_scale=0.9), dt, dur)


def check(cfg, B, init_time=0.0, expect_kernel=None, seed=0):
    pc, bc, tc, dt, dur = cfg
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    if expect_kernel:
        assert eng.last_kernel().startswith(expect_kernel), eng.last_kernel()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, "pos")
    close(vel.cpu().numpy(), rv, "vel", atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
    return eng


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 7, 8, 9, 16])
@pytest.mark.parametrize("mapping", ["1", "2"])
def test_every_dof_class(mp, D, mapping, monkeypatch, mpk_option):
    mpk_option("mapping", mapping)
    for B in (1, 17):
        check(cfg_for(mp, D, 4, 36), B, expect_kernel="k_traj_", seed=D + B)


@pytest.mark.parametrize("mp,nb", [("prodmp", 1), ("prodmp", 2), ("prodmp", 6), ("prodmp", 9), ("prodmp", 13),
                                   ("promp", 1), ("promp", 4), ("promp", 8), ("promp", 15), ("dmp", 1), ("dmp", 7),
                                   ("dmp", 12), ("dmp", 16)])
def test_every_contraction_length(mp, nb):
    """KM = 1..4 k-chunks of the MFMA"""
    eng = check(cfg_for(mp, 3, nb, 40), 9, expect_kernel="k_traj_", seed=nb)
    assert not eng.last_kernel().startswith(("k_traj_rows", "k_traj_phase"))


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("T", [2, 3, 15, 16, 17, 31, 33, 50, 101])
@pytest.mark.parametrize("D", [1, 7])
def test_every_horizon_class(mp, T, D, mpk_option):
    """partial last row tile; T*D % 4 != 0 (unaligned outputs) takes the generic store path"""
    for mapping in ("1", "2"):
        mpk_option("mapping", mapping)
        check(cfg_for(mp, D, 3, T), 5, seed=T)


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
def test_limits_fall_back_to_the_per_episode_kernel(mp):
    # D > 16, shared phase: column groups per episode (dmp with few columns: the per-episode kernels, its Euler loop rules)
    check(cfg_for(mp, 17, 4, 20), 6, expect_kernel="k_traj_wide" if mp != "dmp" else "k_traj_phase")
    check(cfg_for(mp, 2, 20, 20), 6, expect_kernel="k_traj_wide")          # more than 16 basis columns, shared phase
    check(cfg_for(mp, 40, 9, 24), 3, expect_kernel="k_traj_wide" if mp != "dmp" else "k_traj_rows")
    # the same shapes with a per-episode phase: wave per episode, then workgroup per episode (D * KS > 256)
    for D, nb, T, B, kern in ((17, 4, 20, 6, "k_traj_phase"), (40, 9, 24, 3, "k_traj_rows")):
        pc, bc, tc, dt, dur = cfg_for(mp, D, nb, T)
        eng = make_engine(pc, bc, tc, dt, dur)
        params, ip, iv = inputs(pc, bc, tc, B, seed=D)
        p0, v0 = eng.trajectory(params, ip, iv, 0.0)
        assert eng.last_kernel().startswith("k_traj_wide" if mp != "dmp" else kern)
        p1, v1 = eng.trajectory(params, ip, iv, torch.zeros(B, device="cuda"))
        assert eng.last_kernel().startswith(kern), eng.last_kernel()
        if mp != "prodmp" or kern == "k_traj_rows":           # k_traj_phase<prodmp>: c1 / c2 form, not the folded rows
            assert torch.equal(p0, p1) and torch.equal(v0, v1)


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("D,nb,T", [(17, 4, 20), (20, 9, 50), (31, 2, 33), (32, 13, 40), (33, 5, 100), (48, 20, 36),
                                    (64, 3, 17), (100, 6, 24)])
def test_more_than_16_dof_run_as_column_groups_on_the_matrix_cores(mp, D, nb, T):
    """an episode with D > 16 spans ceil(D / 16) column groups of k_traj_wide (the last one ragged); few AND many basis
    columns; ragged last 4-group unit"""
    wide = mp != "dmp" or nb > 16           # dmp, few columns: per-episode kernels (see wide_capable in mpk_host.cpp)
    for B in (1, 7):
        check(cfg_for(mp, D, nb, T), B, init_time=0.25 if mp == "prodmp" else 0.0,
              expect_kernel="k_traj_wide" if wide else "k_traj_", seed=D + B)


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("nb", [14, 17, 29, 64, 130])
@pytest.mark.parametrize("D,T", [(2, 40), (5, 200), (7, 100), (16, 33)])
def test_wide_contractions_run_on_the_matrix_cores(mp, nb, D, T):
    """17+ contraction columns with a shared phase: k-chunked MFMA kernel (k_traj_wide), every MP type, ragged last
    episode group / row tile / k chunk; nb = 14 is wide for prodmp only (14 + goal + 2 boundary columns = 17)"""
    for B in (1, 23):
        eng = check(cfg_for(mp, D, nb, T), B, seed=nb + B)
        wide = nb + {"prodmp": 3, "promp": 1, "dmp": 0}[mp] > 16
        assert eng.last_kernel().startswith("k_traj_wide" if wide else "k_traj_"), eng.last_kernel()
        assert wide or not eng.last_kernel().startswith(("k_traj_rows", "k_traj_wide"))


def test_wide_prodmp_long_horizon_walks_row_tile_blocks():
    """prodmp rows are independent: horizons beyond 16 row tiles run as several row-tile blocks of the same launch"""
    check(cfg_for("prodmp", 3, 20, 300), 7, expect_kernel="k_traj_wide")
    check(cfg_for("prodmp", 7, 40, 523), 5, init_time=0.25, expect_kernel="k_traj_wide")


def test_wide_per_episode_phase_stays_on_the_row_kernel():
    pc, bc, tc, dt, dur = cfg_for("promp", 2, 20, 20)
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, 6, seed=1)
    p0, v0 = eng.trajectory(params, ip, iv, 0.0)
    assert eng.last_kernel().startswith("k_traj_wide")
    p1, v1 = eng.trajectory(params, ip, iv, torch.zeros(6, device="cuda"))
    assert eng.last_kernel().startswith("k_traj_rows")
    assert torch.equal(p0, p1) and torch.equal(v0, v1)      # same rows, same ascending-k fmaf chain: same bits


@pytest.mark.parametrize("B", [1, 64, 1000])
def test_the_reference_example_with_1000_basis_functions(B):
    """
    examples/examples_movement_primitives.py:67 -- `mp_config_override={'basis_generator_kwargs': {'num_basis': 1000}}` on
    fancy_ProMP/Reacher5d-v0 (5 DoF, 200 steps, zero_rbf with one zero-start basis: 5000 parameters per episode).  Round 2
    could not run it at all (the per-episode kernel needs T x K floats of LDS); it is a matrix-core GEMM now.
    """
    cfg = (O.PhaseCfg("linear", tau=4.0), O.BasisCfg("zero_rbf", num_basis=1000, num_basis_zero_start=1,
                                                     num_basis_zero_goal=0, basis_bandwidth_factor=3),
           O.TrajCfg("promp", action_dim=5), 0.02, 4.0)
    eng = check(cfg, B, expect_kernel="k_traj_wide<promp>", seed=B)
    assert eng.num_params == 5000 and eng.num_steps == 200


@pytest.mark.parametrize("flags", [dict(disable_goal=True), dict(disable_weights=True), dict(relative_goal=True),
                                   dict(relative_goal=True, disable_goal=True), dict(auto_scale_basis=True),
                                   dict(auto_scale_basis=True, weights_scale=0.3, goal_scale=2.0, relative_goal=True)])
@pytest.mark.parametrize("init_time", [0.0, 0.3])
def test_prodmp_flags(flags, init_time):
    cfg = cfg_for("prodmp", 5, 4, 60, **flags)
    check(cfg, 11, init_time, expect_kernel="k_traj_")
    # same flags through the per-episode kernel (the reference's c1 / c2 form): against the oracle, and equal to the
    # shared-phase result up to rounding
    pc, bc, tc, dt, dur = cfg
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, 11, seed=3)
    p0, v0 = eng.trajectory(params, ip, iv, init_time)
    p1, v1 = eng.trajectory(params, ip, iv, torch.full((11,), init_time, device="cuda"))
    assert eng.last_kernel().startswith("k_traj_phase")
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    close(p1.cpu().numpy(), rp, "pos"); close(v1.cpu().numpy(), rv, "vel")
    for a, b in ((p0, p1), (v0, v1)):
        a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
        assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()


def test_empty_and_single_episode_batches():
    pc, bc, tc, dt, dur = cfg_for("prodmp", 7, 5, 100)
    eng = make_engine(pc, bc, tc, dt, dur)
    pos, vel = eng.trajectory(np.zeros((0, 42), np.float32), np.zeros((0, 7), np.float32), np.zeros((0, 7), np.float32))
    assert pos.shape == (0, 100, 7) and vel.shape == (0, 100, 7)
    p1, _ = eng.trajectory(np.ones(42, np.float32), np.zeros(7, np.float32), np.zeros(7, np.float32))   # 1-D params
    assert p1.shape == (1, 100, 7)
    with pytest.raises(ValueError):
        eng.trajectory(np.zeros((2, 41), np.float32), np.zeros((2, 7)), np.zeros((2, 7)))


def test_learned_tau_delay_are_clipped_to_their_bounds():
    """np.clip(action, low, high) of black_box_wrapper.py:104-105 happens inside the kernel"""
    pc = O.PhaseCfg("linear", tau=1.0, learn_tau=True, learn_delay=True, tau_bound=(0.3, 0.8), delay_bound=(0.1, 0.2))
    bc, tc = O.BasisCfg("rbf", num_basis=6), O.TrajCfg("promp", action_dim=2)
    eng = make_engine(pc, bc, tc, 0.02, 1.0)
    B = 64
    params, ip, iv = inputs(pc, bc, tc, B, seed=1)
    params[:, 0] = np.linspace(0.05, 1.5, B)        # far outside [0.3, 0.8] on both sides
    params[:, 1] = np.linspace(0.0, 0.5, B)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    rp, rv = O.get_trajectory(pc, bc, tc, params, 1.0, 0.02, 0.0, ip, iv, dtype=np.float64, clip=True)
    close(pos.cpu().numpy(), rp, "clipped pos")
    clipped = np.clip(params, *O.params_bounds(pc, bc, tc))
    assert not np.array_equal(clipped, params)


def test_prodmp_time_beyond_precompute_range_raises():
    """mp_pytorch: RuntimeError('Time is beyond the pre-computation range...')"""
    eng = TrajectoryEngine("prodmp", "exp", "prodmp", 2, 3, dt=0.02, duration=2.0, tau=0.3, basis_alpha=10.0)
    with pytest.raises(RuntimeError, match="pre-computation range"):
        eng.trajectory(np.zeros((4, 8), np.float32), np.zeros((4, 2)), np.zeros((4, 2)), 0.0)     # 2.0 / 0.3 > 6


def test_set_duration_changes_the_horizon():
    pc, bc, tc, dt, dur = cfg_for("promp", 3, 4, 50)
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, 7)
    assert eng.trajectory(params, ip, iv)[0].shape == (7, 50, 3)
    eng.set_duration(0.5, dt)
    assert eng.num_steps == 25
    pos, vel = eng.trajectory(params, ip, iv)
    rp, rv = O.get_trajectory(pc, bc, tc, params, 0.5, dt, 0.0, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, "pos after set_duration")
    assert np.array_equal(eng.times(), O.make_times(0.5, dt, 0.0))


def test_many_init_times_cycle_the_table_cache():
    pc, bc, tc, dt, dur = cfg_for("prodmp", 4, 5, 64)
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, 9)
    first = {}
    for rnd in range(2):
        for k in range(70):                  # more distinct init_times than cache slots (64)
            it = k * dt
            pos, vel = eng.trajectory(params, ip, iv, it)
            if rnd == 0:
                rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, it, ip, iv, dtype=np.float64)
                close(pos.cpu().numpy(), rp, f"init_time {it}")
                first[k] = pos.clone()
            else:
                assert torch.equal(pos, first[k])


@pytest.mark.parametrize("D,T", [(3, 10), (7, 33)])
def test_fused_actions_on_unaligned_shapes(D, T):
    cfg = cfg_for("prodmp", D, 3, T)
    pc, bc, tc, dt, dur = cfg
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 13
    params, ip, iv = inputs(pc, bc, tc, B)
    pg, dg = np.linspace(0.5, 1.5, D), np.linspace(0.05, 0.2, D)
    spec = RolloutSpec("motor", D, pg, dg, -0.7, 0.7, plant="static")
    pos, vel, act = eng.trajectory_actions(params, ip, iv, spec, ip.astype(np.float64), iv.astype(np.float64))
    ra, _, _ = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -0.7, 0.7, "static", dt,
                         ip.astype(np.float64), iv.astype(np.float64))
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
    rp, _ = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, "pos")


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("B", [1, 7, 8, 9, 67])
def test_bulk_input_staging_on_ragged_batches(mp, B, monkeypatch, mpk_option):
    """episode-major kernel with chunked LDS input staging, forced at small sizes: full, ragged and single chunks"""
    mpk_option("mapping", "2")
    mpk_option("bulk", "2")
    mpk_option("quad", "0")
    eng = check(cfg_for(mp, 7, 5, 40), B, expect_kernel="k_traj_stream", seed=B)
    if mp != "dmp":
        cfg = cfg_for(mp, 7, 5, 40)
        pc, bc, tc, dt, dur = cfg
        params, ip, iv = inputs(pc, bc, tc, B, seed=B)
        pg, dg = np.linspace(0.5, 1.5, 7), np.linspace(0.05, 0.2, 7)
        spec = RolloutSpec("motor", 7, pg, dg, -0.7, 0.7, plant="static")
        cp = np.random.default_rng(B).uniform(-1, 1, (B, 7)); cv = np.random.default_rng(B + 1).uniform(-1, 1, (B, 7))
        pos, vel, act = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
        ra, _, _ = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -0.7, 0.7, "static", dt, cp, cv)
        assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
        mpk_option("bulk", "0")
        p2, v2, a2 = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
        assert torch.equal(pos, p2) and torch.equal(vel, v2) and torch.equal(act, a2)


def test_calls_are_capturable_in_a_hip_graph_even_on_a_table_cache_miss():
    """no hipMalloc / synchronisation on the hot path: a replanning sequence (new init_time every plan -> builder kernel
    + trajectory kernel) captured once and replayed"""
    pc, bc, tc, dt, dur = cfg_for("prodmp", 7, 5, 100)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 33
    params, ip, iv = inputs(pc, bc, tc, B, seed=2)
    P, IP, IV = (torch.tensor(x, device="cuda") for x in (params, ip, iv))
    outs = [(torch.empty((B, 100, 7), device="cuda"), torch.empty((B, 100, 7), device="cuda")) for _ in range(4)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(4):
                eng.trajectory(P, IP, IV, 0.5 * k, out=outs[k])         # init_times never used before: cache misses
    torch.cuda.current_stream().wait_stream(side)
    for o in outs:
        o[0].zero_(); o[1].zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in range(4):
        rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.5 * k, ip, iv, dtype=np.float64)
        close(outs[k][0].cpu().numpy(), rp, f"plan {k} pos")
        close(outs[k][1].cpu().numpy(), rv, f"plan {k} vel")


def test_many_captured_ring_graphs_replay(mpk_option):
    """fourteen captured graphs of twenty k_traj_ring launches each (open loop and closed loop, tickets from the device counter),
    alive at once and replayed in turn: with the ticket counter zeroed by a hipMemsetAsync NODE such replays faulted (round 4); the
    counter cleans up after itself now (the last workgroup to leave zeroes it).  Every replay must leave the bits of an eager
    launch.  Round 5: a counter slot belongs to one ordering domain -- here: one capture on one stream, i.e. one graph -- and is
    never handed to another, so graphs replayed CONCURRENTLY (with each other and with eager launches) cannot meet in a counter
    however many are alive (round 4 took slots modulo 256: the 13th graph of 20 launches re-used the first graph's)."""
    pc, bc, tc, dt, dur = cfg_for("prodmp", 7, 5, 100)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 8192
    params, ip, iv = inputs(pc, bc, tc, B, seed=5)
    P, IP, IV = (torch.tensor(x, device="cuda") for x in (params, ip, iv))
    spec = RolloutSpec("motor", 7, np.full(7, 1.2), np.full(7, 0.1), -0.8, 0.8, plant="double_integrator", dt=dt)
    q0 = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (B, 7)), device="cuda")
    cp, cv = q0.clone(), torch.zeros_like(q0)
    mpk_option("ring", 1)
    mpk_option("ring_dbg", 4)                       # closed loop: tickets (its default is b % grid); open loop: b % grid
    ref_closed = [x.clone() for x in eng.trajectory_rollout(P, IP, IV, spec, q0.clone(), torch.zeros_like(q0))]
    assert eng.last_kernel() == "k_traj_ring<prodmp,closed>"
    mpk_option("ring_dbg", 0)                       # open loop: tickets
    spec_s = RolloutSpec("motor", 7, np.full(7, 1.2), np.full(7, 0.1), -0.8, 0.8, plant="static")
    ref_open = [x.clone() for x in eng.trajectory_actions(P, IP, IV, spec_s, cp, cv)]
    assert eng.last_kernel() == "k_traj_ring<prodmp,act>"
    graphs = []
    for i in range(14):
        closed = i % 2 == 0
        mpk_option("ring_dbg", 4 if closed else 0)
        out = tuple(torch.zeros((B, 100, 7), device="cuda") for _ in range(3))
        q, qd = q0.clone(), torch.zeros_like(q0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(20):
                    if closed:
                        q.copy_(q0); qd.zero_()
                        eng.trajectory_rollout(P, IP, IV, spec, q, qd, out=out)
                    else:
                        eng.trajectory_actions(P, IP, IV, spec_s, cp, cv, out=out)
        torch.cuda.current_stream().wait_stream(side)
        graphs.append((g, out, closed))
    torch.cuda.synchronize()
    for rnd in range(25):
        for g, out, closed in (graphs if rnd % 2 == 0 else graphs[::-1]):
            g.replay()
    torch.cuda.synchronize()
    for g, out, closed in graphs:
        for x, y in zip(out, ref_closed if closed else ref_open):
            assert torch.equal(x, y)
    # ... and two at a time on two streams, the first graph against the thirteenth and fourteenth too (the pairs that shared
    # counters in round 4), with eager launches of the same handle on a third stream in between
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for g, out, closed in graphs:
        for x in out:
            x.zero_()
    torch.cuda.synchronize()
    eager = []
    mpk_option("ring_dbg", 0)
    for rnd in range(10):
        pairs = [(i, i + 1) for i in range(0, len(graphs), 2)] + [(0, 12), (1, 13), (0, 13)]
        for i, j in pairs:
            with torch.cuda.stream(sa):
                graphs[i][0].replay()
            with torch.cuda.stream(sb):
                graphs[j][0].replay()
            if rnd < 2:
                with torch.cuda.stream(sc):
                    eager.append(eng.trajectory_actions(P, IP, IV, spec_s, cp, cv))
    torch.cuda.synchronize()
    for g, out, closed in graphs:
        for x, y in zip(out, ref_closed if closed else ref_open):
            assert torch.equal(x, y)
    for out in eager:
        for x, y in zip(out, ref_open):
            assert torch.equal(x, y)


@pytest.mark.parametrize("closed", [False, True])
def test_a_ring_protocol_failure_is_an_error_not_a_wrong_answer(closed, mpk_option):
    """Every spin of k_traj_ring is bounded; a role that gives up leaves outputs unwritten.  Rounds 1 - 4 returned MPK_OK for that
    launch and every later one.  Now the role raises the handle's fault word (mapped host memory): the NEXT entry point on the handle
    and mpk_check_range return MPK_EHIP naming the roles, once; the handle works normally afterwards.  The stall is injected with
    "ring_dbg" 128 (every workgroup's second batch is never published), which counts only with "ablations" 1."""
    from fancy_gym_amd._lib import MPKLibraryError
    pc, bc, tc, dt, dur = cfg_for("prodmp", 7, 5, 100)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 16384
    params, ip, iv = inputs(pc, bc, tc, B, seed=5)
    P, IP, IV = (torch.tensor(x, device="cuda") for x in (params, ip, iv))
    q0 = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (B, 7)), device="cuda")
    spec = RolloutSpec("motor", 7, np.full(7, 1.2), np.full(7, 0.1), -0.8, 0.8, plant="double_integrator" if closed else "static", dt=dt)

    def launch():
        if closed:
            return eng.trajectory_rollout(P, IP, IV, spec, q0.clone(), torch.zeros_like(q0))
        return eng.trajectory_actions(P, IP, IV, spec, q0, torch.zeros_like(q0))
    mpk_option("ring", 1)
    good = [x.clone() for x in launch()]
    assert eng.last_kernel().startswith("k_traj_ring")
    eng.check_range()                                   # nothing pending
    mpk_option("ring_dbg", 128)                         # without "ablations": masked out, the launch is a normal one
    for x, y in zip(launch(), good):
        assert torch.equal(x, y)
    eng.check_range()
    mpk_option("ablations", 1)
    bad = launch()                                      # returns at once (asynchronous); the waves give up ~0.3 s later
    torch.cuda.synchronize()
    assert not all(torch.equal(x, y) for x, y in zip(bad, good)), "the injected stall did not stall anything"
    mpk_option("ring_dbg", 0)
    with pytest.raises(MPKLibraryError, match="gave up waiting") as ei:
        launch()                                        # the next entry point on the handle reports it ...
    # (the WHOLE message: rounds 4 - 5 formatted ~310 characters into char msg[256], and what got cut was the part that matters)
    assert str(ei.value).rstrip().endswith("of that launch are incomplete"), str(ei.value)
    for x, y in zip(launch(), good):                    # ... once: the handle is usable again, and right
        assert torch.equal(x, y)
    # mpk_poll_fault (ABI 4): the same report without a further launch and without synchronising -- the caller has synchronised itself
    mpk_option("ring_dbg", 128)
    launch()
    torch.cuda.synchronize()
    mpk_option("ring_dbg", 0)
    with pytest.raises(MPKLibraryError, match="gave up waiting"):
        eng.poll_fault()
    eng.poll_fault()                                    # reported once
    for x, y in zip(launch(), good):
        assert torch.equal(x, y)
    eng.check_range()
    mpk_option("ring_dbg", 128)
    launch()
    with pytest.raises(MPKLibraryError, match="gave up waiting"):
        eng.check_range()                               # mpk_check_range synchronises, then reports
    mpk_option("ring_dbg", 0)
    eng.check_range()
    for x, y in zip(launch(), good):
        assert torch.equal(x, y)


@pytest.mark.parametrize("quad", ["0", "2", "3", "4"])
@pytest.mark.parametrize("D,T,B", [(7, 200, 1), (7, 200, 9), (3, 33, 21), (16, 40, 5), (1, 50, 70), (5, 17, 4)])
def test_dmp_quad_and_stream_kernels_agree_bitwise_and_match_oracle(quad, D, T, B, monkeypatch, mpk_option):
    mpk_option("dmp_response", 0)          # the serial explicit-Euler kernels (the default route contracts response rows)
    mpk_option("quad", quad)
    eng = check(cfg_for("dmp", D, 5, T), B, seed=T,
                expect_kernel={"0": "k_traj_stream", "2": "k_traj_quad", "3": "k_traj_duo", "4": "k_traj_mono"}[quad])
    pc, bc, tc, dt, dur = cfg_for("dmp", D, 5, T)
    params, ip, iv = inputs(pc, bc, tc, B, seed=T)
    p1, v1 = eng.trajectory(params, ip, iv, 0.0)
    mpk_option("quad", "0")
    p0, v0 = eng.trajectory(params, ip, iv, 0.0)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


@pytest.mark.parametrize("mp", ["prodmp", "promp", "dmp"])
@pytest.mark.parametrize("D,nb,T", [(1, 3, 30), (3, 9, 41), (7, 5, 100), (9, 4, 33), (16, 12, 20), (17, 4, 20),
                                    (32, 3, 12), (64, 1, 7)])
def test_per_episode_kernels_agree_bitwise(mp, D, nb, T, monkeypatch, mpk_option):
    """wave-per-episode (k_traj_phase) and workgroup-per-episode (k_traj_rows) kernels, per-episode init_time: same
    bits from both and from the shared-phase kernels when every episode carries the same init_time (prodmp's
    wave-per-episode kernel uses the reference's c1 / c2 form: equal to rounding)"""
    if mp == "prodmp" and nb + 3 > 16:
        pytest.skip("more than 16 contraction columns")
    pc, bc, tc, dt, dur = cfg_for(mp, D, nb, T)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 37
    params, ip, iv = inputs(pc, bc, tc, B, seed=D * 100 + nb)
    it = torch.full((B,), 0.25 if mp == "prodmp" else 0.0, dtype=torch.float32, device="cuda")
    outs = {}
    if mp == "dmp":
        # bit identity is a property of the EXACT forcing rows; the wave-per-chunk kernel interpolates them by default (round 5:
        # fast_rows_eval, <= 4e-8 of the row apart) -- checked against these bits below
        mpk_option("phase_table", 0)
    for mode in ("1", "0"):
        mpk_option("phase", mode)
        p, v = eng.trajectory(params, ip, iv, it)
        torch.cuda.synchronize()
        outs[mode] = (p.clone(), v.clone(), eng.last_kernel())
    KT = nb + {"prodmp": 3, "promp": 1, "dmp": 3}[mp]      # dmp: goal, y0, ydot0 ride in three extra columns
    KS = 4 if (mp == "promp" and KT <= 4) else (8 if KT <= 8 else 16)
    wave_ok = KT <= 16 and D <= 64 and D * KS <= 256
    assert outs["0"][2].startswith("k_traj_rows")
    if wave_ok:
        assert outs["1"][2].startswith("k_traj_phase"), outs["1"][2]
    def same(a, b):
        if mp != "prodmp" or not outs["1"][2].startswith("k_traj_phase"):
            return torch.equal(a, b)
        a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)      # c1 / c2 form vs folded rows
        return np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
    assert same(outs["0"][0], outs["1"][0]) and same(outs["0"][1], outs["1"][1])
    if D <= 16:
        if mp == "dmp":
            mpk_option("dmp_response", 0)     # the serial explicit-Euler kernels: the per-episode kernels' recurrence, bit for bit
        p0, v0 = eng.trajectory(params, ip, iv, float(it[0]))
        assert not eng.last_kernel().startswith(("k_traj_rows", "k_traj_phase"))
        if mp == "dmp":
            # the per-episode DMP kernels contract their Euler step (five fused operations: dmp_phase_step, round 5), the serial
            # shared-phase kernels round every operation as the reference does: equal to ~1e-7 of the scale, not bit for bit
            for a, b in ((p0, outs["1"][0]), (v0, outs["1"][1])):
                a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
                assert np.abs(a - b).max() <= 2e-6 * max(np.abs(a).max(), 1e-30)
        else:
            assert same(p0, outs["1"][0]) and same(v0, outs["1"][1])
            assert torch.equal(p0, outs["0"][0]) and torch.equal(v0, outs["0"][1])       # folded rows in both
        if mp == "dmp":
            # the default route for a shared phase (response rows on the matrix cores, round 5): equal to rounding
            mpk_option("dmp_response", -1)
            p2, v2 = eng.trajectory(params, ip, iv, float(it[0]))
            if "dmp_resp" in eng.last_kernel():
                for a, b in ((p2, p0), (v2, v0)):
                    a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
                    assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, float(it[0]), ip, iv, dtype=np.float64)
    close(outs["1"][0].cpu().numpy(), rp, "pos")
    close(outs["1"][1].cpu().numpy(), rv, "vel", atol=fd_atol(rp, dt) if mp == "promp" else 0.0)
    if mp == "dmp":
        mpk_option("phase_table", -1); mpk_option("phase", 1)
        p3, v3 = eng.trajectory(params, ip, iv, it)
        close(p3.cpu().numpy(), rp, "pos, interpolated rows"); close(v3.cpu().numpy(), rv, "vel, interpolated rows")
        for a, b in ((p3, outs["1"][0]), (v3, outs["1"][1])):
            a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
            assert np.abs(a - b).max() <= 2e-6 * max(np.abs(b).max(), 1e-30)


def test_batches_beyond_2_31_output_elements():
    """every kernel family with more than 2^31 output elements per array (64-bit addressing); ~60 GB of HBM"""
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2 ** 30:
        pytest.skip("needs 80 GB of free device memory")
    from tools import big_batch_check
    big_batch_check.main()


@pytest.mark.parametrize("phase", ["linear", "exp"])
def test_per_episode_dmp_interpolated_rows_across_the_phase_clip(phase, mpk_option):
    """the per-episode DMP kernels interpolate their forcing rows in a table over the scaled time (one per handle); a LINEAR phase is clipped
    at 1, and the nodes beyond the clip must hold the rows' smooth continuation (items look up min(s, 1)) -- with them clipped too the
    rows within one node of s = 1 were off by 2.5e-6 of the velocity scale (round 5).  Learned tau down to a quarter of the horizon:
    every episode crosses the clip; table against exact rows within 5e-7 of the scale, both kernels, and the oracle"""
    pc = O.PhaseCfg(phase, tau=4.0, alpha_phase=2.0, learn_tau=True, tau_bound=(1.0, 4.0))
    bc = O.BasisCfg("rbf", num_basis=5)
    tc = O.TrajCfg("dmp", action_dim=7, alpha=25.0)
    dt, dur = 0.02, 4.0
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 4000                      # (1 000 chunks of four: one round of the pipeline kernel's resident workgroups)
    params, ip, iv = inputs(pc, bc, tc, B, seed=11)
    params[:, 0] = np.linspace(1.0, 4.0, B, dtype=np.float32)
    for flat in (1, 0):
        mpk_option("phase_flat", flat)
        mpk_option("phase_table", 0)
        p0, v0 = (x.double().clone() for x in eng.trajectory(params, ip, iv, 0.0))
        mpk_option("phase_table", 1)
        p1, v1 = (x.double() for x in eng.trajectory(params, ip, iv, 0.0))
        assert eng.last_kernel() == ("k_traj_phase<dmp,wg,pipe>" if flat else "k_traj_phase<dmp>"), eng.last_kernel()
        ep = float((p1 - p0).abs().max() / p0.abs().max()); ev = float((v1 - v0).abs().max() / v0.abs().max())
        assert ep <= 5e-7 and ev <= 5e-7, (phase, flat, ep, ev)
    n = 64
    rp, rv = O.get_trajectory(pc, bc, tc, params[:n], dur, dt, 0.0, ip[:n], iv[:n], dtype=np.float64)
    close(p1[:n].float().cpu().numpy(), rp, "pos"); close(v1[:n].float().cpu().numpy(), rv, "vel")


@pytest.mark.parametrize("B", [1, 37, 2500, 9001])
def test_prodmp_seven_dof_unrolled_chains_same_bits(B, mpk_option):
    """k_traj_phase<prodmp> at seven DoF and <= 8 columns runs instantiations with the DoF chains unrolled side by side (round 5);
    "pd_generic" 1 takes the run-time DoF loop: the same bits from both, for every row source and round shape, and the oracle"""
    pc = O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.5, 2.0), delay_bound=(0.0, 0.3))
    bc = O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10)
    tc = O.TrajCfg("prodmp", action_dim=7, relative_goal=True)
    dt, dur = 0.02, 2.0
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    for flat in ("0", "1"):
        for table in ("1", "0"):
            mpk_option("phase_table", table); mpk_option("phase_flat", flat)
            mpk_option("pd_generic", "1")
            p0, v0 = (x.clone() for x in eng.trajectory(params, ip, iv, 0.0))
            k0 = eng.last_kernel()
            mpk_option("pd_generic", "0")
            p1, v1 = eng.trajectory(params, ip, iv, 0.0)
            torch.cuda.synchronize()
            assert eng.last_kernel() == k0 and k0.startswith("k_traj_phase<prodmp"), (k0, eng.last_kernel())
            assert torch.equal(p0, p1) and torch.equal(v0, v1), (flat, table, k0)
    n = min(B, 64)
    rp, rv = O.get_trajectory(pc, bc, tc, params[:n], dur, dt, 0.0, ip[:n], iv[:n], dtype=np.float64)
    close(p1[:n].cpu().numpy(), rp, "pos"); close(v1[:n].cpu().numpy(), rv, "vel")


def test_resident_waves_per_cu_are_a_launch_parameter_not_a_result(mpk_option):
    """round 5: how many waves (= write streams) of a kernel a CU holds is chosen by size for k_traj_flat (two workgroups), for
    k_traj_phase<prodmp,lds,flat> (workgroups of four from 24 576 episodes on) and for k_pd_rollout_tiles (four waves beyond 512 MiB);
    "phase_waves" / "tiles_wpb" set it for A/B runs -- whatever the geometry, the same bits"""
    import bench
    from fancy_gym_amd import RolloutSpec
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    B = 30000
    params, ip, iv = inputs(pc, bc, tc, B, seed=4)
    ref = [x.clone() for x in eng.trajectory(params, ip, iv, 0.0)]
    assert eng.last_kernel() == "k_traj_flat<prodmp>", eng.last_kernel()
    for w in (4, 8, 12):
        mpk_option("phase_waves", w)
        p, v = eng.trajectory(params, ip, iv, 0.0)
        assert eng.last_kernel() == "k_traj_flat<prodmp>" and torch.equal(p, ref[0]) and torch.equal(v, ref[1]), w
    mpk_option("phase_waves", -1)
    spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="double_integrator", dt=dt)
    rng = np.random.default_rng(3)
    q0, qd0 = rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))
    acts = []
    for w in (-1, 4, 16, 32):
        mpk_option("phase_waves", w)
        q, qd = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
        acts.append((eng.pd_rollout(spec, ref[0], ref[1], q, qd).clone(), q.clone(), qd.clone()))
    for a_, q_, qd_ in acts[1:]:
        assert torch.equal(a_, acts[0][0]) and torch.equal(q_, acts[0][1]) and torch.equal(qd_, acts[0][2])
    mpk_option("phase_waves", -1)
    pc2 = dataclasses.replace(pc, learn_tau=True, tau_bound=(0.5, 2.0))
    eng2 = make_engine(pc2, bc, tc, dt, duration)
    params2, ip2, iv2 = inputs(pc2, bc, tc, B, seed=6)
    ref2 = [x.clone() for x in eng2.trajectory(params2, ip2, iv2, 0.0)]
    assert eng2.last_kernel() == "k_traj_phase<prodmp,lds,flat>", eng2.last_kernel()
    for key, w in (("tiles_wpb", 2), ("tiles_wpb", 8), ("phase_waves", 4), ("phase_waves", 16)):
        mpk_option(key, w)
        p, v = eng2.trajectory(params2, ip2, iv2, 0.0)
        assert torch.equal(p, ref2[0]) and torch.equal(v, ref2[1]), (key, w)
        mpk_option(key, -1)


@pytest.mark.parametrize("nb,zero_start", [(3, 1), (6, 0)])
@pytest.mark.parametrize("B", [1, 130, 5000])
def test_promp_seven_dof_unrolled_chains_same_bits(B, nb, zero_start, mpk_option):
    """k_traj_phase<promp> at seven DoF (four / eight columns): the DoF chains unrolled side by side against the run-time DoF loop
    ("pd_generic" 1) -- same bits (positions AND the forward-difference velocity across the rounds of 63 steps), and the oracle"""
    dt, dur = 0.008, 2.8
    pc = O.PhaseCfg("linear", tau=dur, learn_tau=True, learn_delay=True, tau_bound=(0.5, dur), delay_bound=(0.05, 0.15))
    bc = O.BasisCfg("zero_rbf" if zero_start else "rbf", num_basis=nb, num_basis_zero_start=zero_start, num_basis_zero_goal=zero_start)
    tc = O.TrajCfg("promp", action_dim=7)
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + nb)
    mpk_option("pd_generic", "1")
    p0, v0 = (x.clone() for x in eng.trajectory(params, ip, iv, 0.0))
    assert eng.last_kernel() == "k_traj_phase<promp>", eng.last_kernel()
    mpk_option("pd_generic", "0")
    p1, v1 = eng.trajectory(params, ip, iv, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(v0, v1)
    n = min(B, 32)
    rp, rv = O.get_trajectory(pc, bc, tc, params[:n], dur, dt, 0.0, ip[:n], iv[:n], dtype=np.float64)
    close(p1[:n].cpu().numpy(), rp, "pos"); close(v1[:n].cpu().numpy(), rv, "vel", atol=fd_atol(rp, dt))


@pytest.mark.parametrize("name", ["prodmp_learn_tau_delay", "cfg2_learn_tau"])
def test_prodmp_row_table_in_lds_or_l2_same_bits(name, monkeypatch, mpk_option):
    """k_traj_phase<prodmp> gathers its fp32 rows from an LDS copy of the table when that fits beside 8 waves, else
    from L2: same arithmetic, same bits; both against the oracle"""
    from tests.test_gpu_trajectory import PER_ROW
    if name == "cfg2_learn_tau":
        cfg = (O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0, learn_tau=True, tau_bound=(0.5, 2.0)),
               O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10), O.TrajCfg("prodmp", action_dim=7),
               0.02, 2.0)
    else:
        cfg = PER_ROW[name]
    pc, bc, tc, dt, dur = cfg
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 2500
    params, ip, iv = inputs(pc, bc, tc, B, seed=2)
    outs = {}
    for flat in ("0", "1"):
        for mode in ("1", "0"):
            mpk_option("phase_table", mode); mpk_option("phase_flat", flat)
            p, v = eng.trajectory(params, ip, iv, 0.0)
            torch.cuda.synchronize()
            outs[mode + flat] = (p.clone(), v.clone(), eng.last_kernel())
    assert outs["00"][2] == "k_traj_phase<prodmp>" and outs["01"][2] == "k_traj_phase<prodmp,flat>"
    assert outs["10"][2] == "k_traj_phase<prodmp,lds>" and outs["11"][2] == "k_traj_phase<prodmp,lds,flat>", outs["11"][2]
    for k in ("10", "01", "11"):
        assert torch.equal(outs["00"][0], outs[k][0]) and torch.equal(outs["00"][1], outs[k][1]), k
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    close(outs["11"][0].cpu().numpy(), rp, "pos"); close(outs["11"][1].cpu().numpy(), rv, "vel")


@pytest.mark.parametrize("T", [2, 17, 63, 64, 65, 100, 129, 350])
@pytest.mark.parametrize("D,nb", [(1, 2), (7, 5), (9, 12), (16, 3)])
def test_prodmp_flat_rounds_over_a_chunk_same_bits(T, D, nb, mpk_option):
    """k_traj_phase<prodmp, flat>: 64 consecutive (episode, step) items of a chunk per round instead of 64 steps of one
    episode -- per-lane tau / delay / init_time / boundary factors; every chunk size up to 8, ragged last chunk, learned
    tau + delay with per-episode init_time: same bits as the per-episode rounds, and the oracle"""
    dt = 0.01
    dur = T * dt
    pc = O.PhaseCfg("exp", tau=dur, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.5 * dur, 1.2 * dur),
                    delay_bound=(0.0, 0.1 * dur))
    bc = O.BasisCfg("prodmp", num_basis=nb, alpha=15)
    tc = O.TrajCfg("prodmp", action_dim=D, weights_scale=0.8, goal_scale=1.1, relative_goal=bool(D & 1))
    eng = make_engine(pc, bc, tc, dt, dur)
    for B in (1, 3, 37):
        params, ip, iv = inputs(pc, bc, tc, B, seed=B + T)
        rng = np.random.default_rng(T + D)
        params[:, 0] = rng.uniform(0.4 * dur, 1.3 * dur, B)
        params[:, 1] = rng.uniform(-0.02 * dur, 0.12 * dur, B)
        it = torch.tensor(rng.integers(0, 3, B) * dt, dtype=torch.float32, device="cuda")
        mpk_option("phase_flat", "0"); mpk_option("phase_chunk", "-1")
        p0, v0 = (x.clone() for x in eng.trajectory(params, ip, iv, it))
        assert eng.last_kernel().startswith("k_traj_phase<prodmp") and "flat" not in eng.last_kernel()
        mpk_option("phase_flat", "1")
        for chunk in ("-1", "1", "2", "3", "5", "8"):
            mpk_option("phase_chunk", chunk)
            p1, v1 = eng.trajectory(params, ip, iv, it)
            torch.cuda.synchronize()
            assert eng.last_kernel().endswith("flat>"), eng.last_kernel()
            assert torch.equal(p0, p1) and torch.equal(v0, v1), (B, chunk)
        mpk_option("phase_chunk", "-1")
        for b in range(min(B, 3)):
            rp, rv = O.get_trajectory(pc, bc, tc, params[b:b + 1], dur, dt, float(it[b]), ip[b:b + 1], iv[b:b + 1],
                                      dtype=np.float64)
            close(p0[b:b + 1].cpu().numpy(), rp, "pos"); close(v0[b:b + 1].cpu().numpy(), rv, "vel")


@pytest.mark.parametrize("name", ["prodmp_learn_tau_delay", "promp_learn_tau", "dmp_learn_delay"])
@pytest.mark.parametrize("B", [1, 7, 130])
def test_per_episode_chunking_is_invisible(name, B, monkeypatch, mpk_option):
    """a wave's episodes are worked on in chunks of 1, 2 or 4 consecutive episodes (ragged last chunk; dmp: one lane per
    (episode, DoF) of the chunk in the Euler recurrence, up to 16 episodes): same bits"""
    from tests.test_gpu_trajectory import PER_ROW
    pc, bc, tc, dt, dur = PER_ROW[name]
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    it = torch.tensor(np.random.default_rng(B).integers(0, 4, B) * dt, dtype=torch.float32, device="cuda")
    ref = None
    for chunk in ("1", "2", "4") + (("3", "9", "16") if name.startswith("dmp") else ()):
        mpk_option("phase_chunk", chunk)
        p, v = eng.trajectory(params, ip, iv, it)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith("k_traj_phase")
        if ref is None:
            ref = (p.clone(), v.clone())
        assert torch.equal(p, ref[0]) and torch.equal(v, ref[1]), chunk
    for b in range(min(B, 5)):
        rp, rv = O.get_trajectory(pc, bc, tc, params[b:b + 1], dur, dt, float(it[b]), ip[b:b + 1], iv[b:b + 1],
                                  dtype=np.float64)
        close(ref[0][b:b + 1].cpu().numpy(), rp, "pos")


def test_per_episode_prodmp_range_error_is_reported():
    """learned tau small enough that the plan leaves the pre-computed table (scaled time > 6): the single-episode path
    raises like mp_pytorch; the batched path clamps and reports through check_range()"""
    pc = O.PhaseCfg("exp", tau=1.0, alpha_phase=3.0, learn_tau=True, tau_bound=(0.01, 2.0))
    bc = O.BasisCfg("prodmp", num_basis=4, alpha=10)
    tc = O.TrajCfg("prodmp", action_dim=3)
    eng = make_engine(pc, bc, tc, 0.02, 1.0)
    params, ip, iv = inputs(pc, bc, tc, 4, seed=1)
    params[:, 0] = [1.0, 0.9, 0.1, 1.2]          # duration / 0.1 = 10 > 6
    eng.trajectory(params, ip, iv, torch.zeros(4, device="cuda"))
    with pytest.raises(RuntimeError, match="pre-computation range"):
        eng.check_range()
    eng.check_range()                             # the flag is cleared by the check
    params[:, 0] = 1.0
    eng.trajectory(params, ip, iv, torch.zeros(4, device="cuda"))
    eng.check_range()
    with pytest.raises(RuntimeError, match="pre-computation range"):
        eng.trajectory_host(np.concatenate([[0.1], params[0, 1:]]).astype(np.float32), ip[0], iv[0], 0.0)
    # the batched siblings of BlackBoxWrapper raise the same error instead of returning clamped trajectories
    from fancy_gym_amd import BatchedBlackBox
    from fancy_gym_amd.black_box.factory import (get_basis_generator, get_controller, get_phase_generator,
                                                 get_trajectory_generator)
    pg = get_phase_generator("exp", tau=1.0, alpha_phase=3.0, learn_tau=True, tau_bound=[0.01, 2.0])
    tg = get_trajectory_generator("prodmp", 3, get_basis_generator("prodmp", pg, num_basis=4, alpha=10))
    bb = BatchedBlackBox(tg, get_controller("motor", p_gains=1.0, d_gains=0.1), 4, 0.02, 1.0, plant="double_integrator")
    bb.reset()
    bad = params.copy(); bad[:, 0] = [1.0, 0.9, 0.1, 1.2]
    bb.step(bad)
    with pytest.raises(RuntimeError, match="pre-computation range"):
        bb.reset()                                # the episode just finished left the table range
    bb.reset()
    bb.step(params)
    bb.check_range()
    bb.reset()
    pos, vel = eng.trajectory_host(params[0], ip[0], iv[0], 0.0)
    rp, rv = O.get_trajectory(pc, bc, tc, params[:1], 1.0, 0.02, 0.0, ip[:1], iv[:1], dtype=np.float64)
    assert not pos.is_cuda and pos.shape == (50, 3)
    close(pos.numpy()[None], rp, "pos"); close(vel.numpy()[None], rv, "vel")


# ---- canaries: no kernel writes outside its output arrays ---------------------------------------------------------------
GUARD = 2048
SENTINEL = 0x7FC0DEAD


def _guarded(shape, dtype, offset=0):
    """a tensor view with GUARD sentinel elements before and after it (offset: extra elements of misalignment)"""
    n = int(np.prod(shape))
    esize = torch.tensor([], dtype=dtype).element_size()
    raw = torch.full((2 * GUARD + n + offset + 8,), 0, dtype=dtype, device="cuda")
    bits = raw.view(torch.int32 if esize == 4 else torch.int64)
    bits.fill_(SENTINEL)
    view = raw[GUARD + offset:GUARD + offset + n].view(shape)
    return raw, view, (GUARD + offset, GUARD + offset + n)


def _intact(raw, span):
    bits = raw.view(torch.int32 if raw.element_size() == 4 else torch.int64)
    return bool((bits[:span[0]] == SENTINEL).all()) and bool((bits[span[1]:] == SENTINEL).all())


@pytest.mark.parametrize("name", ["cfg2", "cfg5", "cfg3", "prodmp_learn_tau_delay", "promp_learn_tau", "dmp_learn_delay"])
@pytest.mark.parametrize("B", [1, 3, 37, 131, 2100])
@pytest.mark.parametrize("offset", [0, 1])
def test_no_writes_outside_the_output_arrays(name, B, offset, monkeypatch, mpk_option):
    """every trajectory kernel family, ragged batches, 16-byte aligned and misaligned outputs: guard words before and
    after pos / vel / actions stay untouched (the GPU address sanitizer is not available on this pool)"""
    from tests.test_gpu_trajectory import CFG2, CFG3, CFG5, PER_ROW
    cfg = {"cfg2": CFG2, "cfg5": CFG5, "cfg3": CFG3}.get(name) or PER_ROW[name]
    pc, bc, tc, dt, dur = cfg
    eng = make_engine(pc, bc, tc, dt, dur)
    T, D = eng.num_steps, eng.num_dof
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    # inputs sit between NaN pads (a kernel that reads past an input array and USES the value poisons its output), and with
    # offset = 1 they do not start on a 16-byte boundary either
    def padded(x):
        raw = torch.full((x.size + 256 + offset,), float("nan"), dtype=torch.float32, device="cuda")
        v = raw[128 + offset:128 + offset + x.size].view(x.shape)
        v.copy_(torch.tensor(x))
        return v
    params, ip, iv = padded(params), padded(ip), padded(iv)
    assert params.data_ptr() % 16 == 4 * offset
    for mapping in ("1", "2"):
        mpk_option("mapping", mapping)
        (rp, pos, sp), (rv, vel, sv) = _guarded((B, T, D), torch.float32, offset), _guarded((B, T, D), torch.float32, offset)
        eng.trajectory(params, ip, iv, 0.0, out=(pos, vel))
        torch.cuda.synchronize()
        assert _intact(rp, sp) and _intact(rv, sv), (eng.last_kernel(), "trajectory")
        assert bool(torch.isfinite(pos).all()) and bool(torch.isfinite(vel).all()), (eng.last_kernel(), "poisoned")
        want = eng.trajectory(params, ip, iv, 0.0)
        assert torch.equal(pos, want[0]) and torch.equal(vel, want[1])
    mpk_option("mapping", -1)
    nchk = min(B, 4)
    host = [np.asarray(torch.as_tensor(x).cpu()) for x in (params, ip, iv)]
    rp, rv = O.get_trajectory(pc, bc, tc, host[0][:nchk], dur, dt, 0.0, host[1][:nchk], host[2][:nchk], dtype=np.float64)
    close(pos[:nchk].cpu().numpy(), rp, "pos")
    close(vel[:nchk].cpu().numpy(), rv, "vel",
          atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
    # fused / two-launch actions and the closed-loop rollout through the same guarded buffers
    spec_s = RolloutSpec("motor", D, 1.0, 0.1, -1.0, 1.0, plant="static")
    spec_d = RolloutSpec("motor", D, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=dt)
    cp = torch.as_tensor(ip, device="cuda").double().contiguous(); cv = torch.zeros_like(cp)
    for quad in ("0", "2"):
        mpk_option("quad", quad); mpk_option("pd_quad", quad)
        mpk_option("bulk", "2" if quad == "2" else "1")
        bufs = [_guarded((B, T, D), torch.float32, offset) for _ in range(3)]
        eng.trajectory_actions(params, ip, iv, spec_s, cp, cv, out=tuple(b[1] for b in bufs))
        torch.cuda.synchronize()
        assert all(_intact(b[0], b[2]) for b in bufs), (eng.last_kernel(), "actions")
        assert all(bool(torch.isfinite(b[1]).all()) for b in bufs), (eng.last_kernel(), "poisoned actions")
        bufs = [_guarded((B, T, D), torch.float32, offset) for _ in range(3)]
        rq, q, sq = _guarded((B, D), torch.float64); rqd, qd, sqd = _guarded((B, D), torch.float64)
        q.copy_(cp); qd.copy_(cv)
        n_steps = torch.tensor(np.random.default_rng(B).integers(0, T + 1, B).astype(np.int32), device="cuda")
        eng.trajectory_rollout(params, ip, iv, spec_d, q, qd, n_steps=n_steps, out=tuple(b[1] for b in bufs))
        torch.cuda.synchronize()
        assert all(_intact(b[0], b[2]) for b in bufs) and _intact(rq, sq) and _intact(rqd, sqd), (eng.last_kernel(), "rollout")
        assert all(bool(torch.isfinite(b[1]).all()) for b in bufs) and bool(torch.isfinite(q).all())


@pytest.mark.parametrize("D,T", [(5, 200), (7, 100), (3, 10), (16, 40)])
@pytest.mark.parametrize("B", [1, 9, 131])
@pytest.mark.parametrize("quad", ["0", "2"])
def test_no_writes_outside_the_rollout_outputs(D, T, B, quad, monkeypatch, mpk_option):
    mpk_option("pd_quad", quad)
    eng = TrajectoryEngine(device=0, mp_type="promp", phase_type="linear", basis_type="rbf", num_dof=D, num_basis=3,
                           dt=0.01, duration=T * 0.01, tau=T * 0.01)
    rng = np.random.default_rng(B)
    def padded(x):      # NaN on both sides: a read past the array that is USED would poison the outputs
        raw = torch.full((x.size + 512,), float("nan"), dtype=torch.float32, device="cuda")
        v = raw[256:256 + x.size].view(x.shape)
        v.copy_(torch.tensor(x))
        return v
    des_pos = padded(rng.standard_normal((B, T, D)).astype(np.float32))
    des_vel = padded(rng.standard_normal((B, T, D)).astype(np.float32))
    spec = RolloutSpec("motor", D, 0.6, 0.075, -2.0, 2.0, plant="double_integrator", dt=0.01)
    n_steps = torch.tensor(rng.integers(0, T + 1, B).astype(np.int32), device="cuda")
    ra, act, sa = _guarded((B, T, D), torch.float32)
    rr, rew, sr = _guarded((B, T), torch.float64)
    rq, q, sq = _guarded((B, D), torch.float64); rqd, qd, sqd = _guarded((B, D), torch.float64)
    q.zero_(); qd.zero_()
    goal = torch.zeros((B, 2), dtype=torch.float64, device="cuda")
    lib, h = eng._lib, eng._h
    import ctypes as C
    from fancy_gym_amd import _lib
    _lib.check(lib.mpk_reacher_rollout(h, C.byref(spec.c), des_pos.data_ptr(), des_vel.data_ptr(), q.data_ptr(),
                                       qd.data_ptr(), n_steps.data_ptr(), None, goal.data_ptr(), 199, act.data_ptr(),
                                       rew.data_ptr(), B, T, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert _intact(ra, sa) and _intact(rr, sr) and _intact(rq, sq) and _intact(rqd, sqd)
    assert bool(torch.isfinite(act).all()) and bool(torch.isfinite(rew).all()) and bool(torch.isfinite(q).all())
    ra, act, sa = _guarded((B, T, D), torch.float32)
    _lib.check(lib.mpk_pd_rollout(h, C.byref(spec.c), des_pos.data_ptr(), des_vel.data_ptr(), q.data_ptr(),
                                  qd.data_ptr(), n_steps.data_ptr(), act.data_ptr(), B, T,
                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert _intact(ra, sa) and _intact(rq, sq) and _intact(rqd, sqd)


@pytest.mark.parametrize("B", [1, 5, 257, 1030])
def test_no_writes_outside_the_small_outputs(B):
    """integer replanning state, validity / penalty reduction, condition gather"""
    from tests.test_gpu_trajectory import CFG2
    pc, bc, tc, dt, dur = CFG2
    eng = make_engine(pc, bc, tc, dt, dur)
    T, D = eng.num_steps, eng.num_dof
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    import ctypes as C
    from fancy_gym_amd import _lib
    lib, h, st = eng._lib, eng._h, torch.cuda.current_stream().cuda_stream
    rt, ts, s1 = _guarded((B,), torch.int32); rp, ps, s2 = _guarded((B,), torch.int32)
    rs, seg, s3 = _guarded((B,), torch.int32)
    rd = torch.full((B + 2 * GUARD,), 0x5A, dtype=torch.uint8, device="cuda"); done = rd[GUARD:GUARD + B]
    ts.zero_(); ps.zero_(); done.zero_()
    _lib.check(lib.mpk_replan_advance(h, ts.data_ptr(), ps.data_ptr(), seg.data_ptr(), done.data_ptr(), 25, 4, 100, T, B, st))
    rv = torch.full((B + 2 * GUARD,), 0x5A, dtype=torch.uint8, device="cuda"); valid = rv[GUARD:GUARD + B]
    rpen, pen, s4 = _guarded((B,), torch.float64)
    lo = (C.c_double * D)(*([-0.5] * D)); hi = (C.c_double * D)(*([0.5] * D))
    _lib.check(lib.mpk_traj_validity_penalty(h, pos.data_ptr(), None, C.cast(lo, C.c_void_p), C.cast(hi, C.c_void_p), 0,
                                             None, None, valid.data_ptr(), pen.data_ptr(), B, T, st))
    rc, cpos, s5 = _guarded((B, D), torch.float32); rcv, cvel, s6 = _guarded((B, D), torch.float32)
    _lib.check(lib.mpk_condition_gather(h, pos.data_ptr(), vel.data_ptr(), seg.data_ptr(), cpos.data_ptr(),
                                        cvel.data_ptr(), B, T, st))
    torch.cuda.synchronize()
    assert _intact(rt, s1) and _intact(rp, s2) and _intact(rs, s3) and _intact(rpen, s4) and _intact(rc, s5) and _intact(rcv, s6)
    for raw in (rd, rv):
        assert bool((raw[:GUARD] == 0x5A).all()) and bool((raw[GUARD + B:] == 0x5A).all())
    assert bool((seg == 25).all()) and torch.equal(cpos, pos[:, 24]) and torch.equal(cvel, vel[:, 24])


@pytest.mark.parametrize("name", ["cfg2", "cfg5", "cfg3", "prodmp_learn_tau_delay", "promp_learn_tau"])
def test_repeated_launches_give_identical_bits(name):
    """LDS staging, cross-lane exchanges and prefetch double buffers leave no run-to-run variation: ten launches of every
    kernel family at a batch that fills the chip several times over"""
    from tests.test_gpu_trajectory import CFG2, CFG3, CFG5, PER_ROW
    pc, bc, tc, dt, dur = {"cfg2": CFG2, "cfg5": CFG5, "cfg3": CFG3}.get(name) or PER_ROW[name]
    eng = make_engine(pc, bc, tc, dt, dur)
    D = eng.num_dof
    B = 40000
    params, ip, iv = inputs(pc, bc, tc, B, seed=3)
    params, ip, iv = (torch.tensor(x, device="cuda") for x in (params, ip, iv))
    spec = RolloutSpec("motor", D, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=dt)
    first = None
    for rep in range(10):
        q, qd = ip.double().contiguous(), iv.double().contiguous()
        pos, vel, act = eng.trajectory_rollout(params, ip, iv, spec, q, qd)
        p2, v2 = eng.trajectory(params, ip, iv, 0.0)
        torch.cuda.synchronize()
        cur = (pos.clone(), vel.clone(), act.clone(), q.clone(), qd.clone(), p2.clone(), v2.clone())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, cur)), rep


@pytest.mark.parametrize("mp,T", [("prodmp", 1000), ("prodmp", 8000), ("promp", 3000), ("dmp", 1000), ("dmp", 3000)])
def test_long_horizons_take_the_kernels_that_fit(mp, T):
    """horizons whose basis tables outgrow the episode-major kernel's LDS: tile-major / per-episode kernels for the
    trajectory, trajectory + rollout launches behind mpk_trajectory_rollout -- same answers"""
    pc, bc, tc, dt, dur = cfg_for(mp, 3, 4, T, dt=0.002)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 5
    params, ip, iv = inputs(pc, bc, tc, B, seed=T)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, "pos")
    close(vel.cpu().numpy(), rv, "vel", atol=fd_atol(rp, dt) if mp == "promp" else 0.0)
    p2, v2 = eng.trajectory(params, ip, iv, torch.zeros(B, device="cuda"))
    close(p2.cpu().numpy(), rp, "per-episode pos")
    spec = RolloutSpec("motor", 3, 1.0, 0.1, -1.0, 1.0, plant="double_integrator", dt=dt)
    q0 = ip.astype(np.float64)
    q, qd = torch.tensor(q0, device="cuda"), torch.zeros((B, 3), dtype=torch.float64, device="cuda")
    p3, v3, act = eng.trajectory_rollout(params, ip, iv, spec, q, qd)
    assert torch.equal(p3, pos) and torch.equal(v3, vel)
    ra, rq, rqd = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", 1.0, 0.1, -1.0, 1.0, "double_integrator", dt,
                            q0, np.zeros((B, 3)))
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32)) and np.array_equal(q.cpu().numpy(), rq)


@pytest.mark.parametrize("D,nb,T,B", [(7, 5, 200, 1), (7, 5, 200, 37), (7, 5, 200, 1026), (3, 4, 70, 9), (16, 9, 33, 50), (5, 2, 64, 130),
                                      (7, 5, 200, 6003), (8, 3, 50, 9001), (5, 5, 100, 7000),      # (these three: chunks of eight episodes)
                                      (7, 5, 200, 4096), (7, 5, 200, 10240)])   # cfg3 + learned tau where the kernel is automatic: chunks of four / eight
def test_per_episode_dmp_workgroup_and_wave_kernels_agree_bitwise(D, nb, T, B, mpk_option):
    """learned tau (per-episode phase) DMP: the workgroup-per-chunk kernel (four tiles of rows at once, the automatic choice for a few
    thousand episodes) and the wave-per-chunk kernel leave the same bits, and both follow the oracle"""
    pc, bc, tc, dt, dur = cfg_for("dmp", D, nb, T)
    pc = dataclasses.replace(pc, learn_tau=True, tau_bound=(0.5 * dur, 1.5 * dur))
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + T)
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    # forcing rows exact (0) / interpolated from the handle's table (1: round 5) in ALL THREE kernels: the same bits from each; the
    # automatic setting against the oracle
    pipe_seen = set()
    for table in (0, 1):
        mpk_option("phase_table", table)
        mpk_option("phase_flat", 1)
        mpk_option("pipe", 0)                           # the plain workgroup kernel: rows -> Euler steps -> stores, block after block
        p1, v1 = [x.clone() for x in eng.trajectory(params, ip, iv, 0.0)]
        assert eng.last_kernel() == "k_traj_phase<dmp,wg>", eng.last_kernel()
        mpk_option("pipe", -1)                          # ... and as a pipeline (round 5) where it applies: <= 4 episodes per chunk, 8 columns,
        p2, v2 = [x.clone() for x in eng.trajectory(params, ip, iv, 0.0)]     # T > 48, one round of resident workgroups
        assert eng.last_kernel() in ("k_traj_phase<dmp,wg>", "k_traj_phase<dmp,wg,pipe>"), eng.last_kernel()
        pipe_seen.add(eng.last_kernel())
        assert torch.equal(p1, p2) and torch.equal(v1, v2), (table, eng.last_kernel())
        mpk_option("phase_flat", 0)
        p0, v0 = eng.trajectory(params, ip, iv, 0.0)
        assert eng.last_kernel() == "k_traj_phase<dmp>", eng.last_kernel()
        assert torch.equal(p1, p0) and torch.equal(v1, v0), table
        close(p1.cpu().numpy(), rp, "pos"); close(v1.cpu().numpy(), rv, "vel")
    if (D, nb, T) == (7, 5, 200) and B <= 4096:
        assert "k_traj_phase<dmp,wg,pipe>" in pipe_seen, pipe_seen
    mpk_option("phase_table", -1); mpk_option("phase_flat", -1)
    p2, v2 = eng.trajectory(params, ip, iv, 0.0)
    close(p2.cpu().numpy(), rp, "pos (automatic)"); close(v2.cpu().numpy(), rv, "vel (automatic)")


def test_dmp_horizon_beyond_the_lds_of_the_shared_phase_kernels_runs_time_tiled():
    """T = 8000: the forcing / state rows of a whole horizon no longer fit any kernel's LDS (round 1 refused this
    shape); the per-episode DMP kernel walks the horizon in 16-step tiles and takes it"""
    pc, bc, tc, dt, dur = cfg_for("dmp", 3, 4, 8000, dt=0.002)
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, 2, seed=1)
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    _lib.set_option("dmp_response", 0)
    try:
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        assert eng.last_kernel().startswith("k_traj_phase<dmp"), eng.last_kernel()
        close(pos.cpu().numpy(), rp, "pos"); close(vel.cpu().numpy(), rv, "vel")
    finally:
        _lib.reset_options()
    # round 5: the response route has no such limit (the tile-major kernel reads its rows from the table in memory)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    assert eng.last_kernel() == "k_traj_tiles<dmp_resp>", eng.last_kernel()
    close(pos.cpu().numpy(), rp, "pos"); close(vel.cpu().numpy(), rv, "vel")


def test_reciprocal_division_is_the_ieee_quotient_for_every_grid_time():
    """
    The table indices of the per-episode-phase ProDMP kernel -- the bit-exact integer part of the path -- come from
    q = fma(fma(-d, z*r, z), r, z*r) with r = 1/d taken once (mpk.h: mpk_selftest_division) instead of two IEEE divisions
    per step.  Exhaustive over the numerators a BASELINE time grid can produce: every fp32 in [2^-20, 16) (t - delay for
    horizons up to 16 s; scaled times up to the 6-tau table range) and zero, for 64 divisors: the reference's tau bounds
    and defaults, their fp32 neighbours, the grid steps basis_dt / tau0, random taus -- and the all-ones significand that
    the correction step cannot handle and that therefore takes the IEEE division.
    """
    import ctypes as C
    from fancy_gym_amd import _lib
    pc, bc, tc, dt, dur = CFG2
    eng = make_engine(pc, bc, tc, dt, dur)
    lib = _lib.load()
    rng = np.random.default_rng(0)
    f32 = np.float32
    divisors = [0.8, 1.5, 2.8, 4.0, 0.3, 2.0, 1e-5, 0.04, 0.5, 1.0, 3.0, 0.05, 0.15,
                float(f32(0.01) / f32(1.5)), float(f32(0.01) / f32(2.8)), float(f32(0.01) / f32(4.0)), float(f32(0.01) / f32(1.0))]
    divisors += [float(np.nextafter(f32(d), f32(np.inf))) for d in (0.8, 1.5)] + [float(np.nextafter(f32(d), f32(0))) for d in (0.8, 1.5)]
    divisors += [float(f32(np.frombuffer(np.uint32(0x3fffffff).tobytes(), np.float32)[0]))]      # significand all ones
    divisors += [float(f32(x)) for x in rng.uniform(0.05, 4.0, 64 - len(divisors))]
    lo = int(np.frombuffer(f32(2.0 ** -20).tobytes(), np.uint32)[0])
    hi = int(np.frombuffer(f32(16.0).tobytes(), np.uint32)[0])
    bad = C.c_uint64()
    total = 0
    for d in divisors:
        _lib.check(lib.mpk_selftest_division(eng._h, f32(d), lo, hi - lo, C.byref(bad), None))
        assert bad.value == 0, (d, bad.value)
        _lib.check(lib.mpk_selftest_division(eng._h, f32(d), 0, 1, C.byref(bad), None))      # zero
        assert bad.value == 0
        total += hi - lo
    assert total > 3e9


@pytest.mark.parametrize("mp_type", ["promp", "dmp", "prodmp"])
def test_show_scaled_basis_matches_the_oracle_basis(mp_type):
    """traj_gen.show_scaled_basis (examples/mp_params_tuning.py:7): 1000 times from delay - tau to delay + 2 tau, basis x
    scale, from the device row functions"""
    from fancy_gym_amd.black_box.factory import get_basis_generator, get_phase_generator, get_trajectory_generator
    tau = 1.5
    if mp_type == "prodmp":
        pg = get_phase_generator("exp", tau=tau, alpha_phase=3.0)
        bg = get_basis_generator("prodmp", pg, num_basis=5, alpha=10, basis_bandwidth_factor=2)
        tg = get_trajectory_generator("prodmp", 3, bg, weights_scale=0.7, goal_scale=0.4, auto_scale_basis=True)
        pc = O.PhaseCfg("exp", tau=tau, alpha_phase=3.0)
        bc = O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10)
        tc = O.TrajCfg("prodmp", action_dim=3, weights_scale=0.7, goal_scale=0.4, auto_scale_basis=True)
    elif mp_type == "promp":
        pg = get_phase_generator("linear", tau=tau)
        bg = get_basis_generator("zero_rbf", pg, num_basis=5, num_basis_zero_start=1, num_basis_zero_goal=0)
        tg = get_trajectory_generator("promp", 3, bg, weights_scale=2.0)
        pc = O.PhaseCfg("linear", tau=tau)
        bc = O.BasisCfg("zero_rbf", num_basis=5, num_basis_zero_start=1, num_basis_zero_goal=0)
        tc = O.TrajCfg("promp", action_dim=3, weights_scale=2.0)
    else:
        pg = get_phase_generator("exp", tau=tau, alpha_phase=2.0)
        bg = get_basis_generator("rbf", pg, num_basis=4)
        tg = get_trajectory_generator("dmp", 3, bg, weights_scale=3.0)
        pc = O.PhaseCfg("exp", tau=tau, alpha_phase=2.0)
        bc = O.BasisCfg("rbf", num_basis=4)
        tc = O.TrajCfg("dmp", action_dim=3, weights_scale=3.0)
    times, basis = tg.show_scaled_basis()
    assert times.shape == (1000,) and times[0] == np.float32(-tau) and times[-1] == np.float32(2 * tau)
    if mp_type == "prodmp":
        tabs = O.prodmp_tables(pc, bc, np.float64)
        idx = O.prodmp_indices(times, np.float32(tau), np.float32(0.0), tabs.scaled_dt)
        ref = tabs.pos_basis[idx] * O.prodmp_weights_goal_scale(tc, bc, O.prodmp_tables(pc, bc, np.float32), np.float32)
        assert basis.shape == (1000, 6)
    else:
        phi = O.rbf_basis(pc, bc, times.astype(np.float64), dtype=np.float64)
        zs = bc.num_basis_zero_start if bc.basis_generator_type == "zero_rbf" else 0
        ref = phi[:, zs:zs + bc.num_basis] * tc.weights_scale
        assert basis.shape == (1000, bc.num_basis)
    close(basis, ref, "scaled basis")
    assert np.all(basis[times < 0] == basis[0]) or mp_type != "prodmp"      # before the delay: table index 0


_TAIL_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np, torch
from oracle import mp_oracle as O
from tests.test_gpu_trajectory import inputs, make_engine
from fancy_gym_amd import _lib
torch.cuda.set_device(0)
for flags in (dict(disable_goal=True), dict(disable_weights=True)):
    for D, nb, T, B in ((4, 6, 54, 16), (16, 2, 27, 100), (8, 6, 66, 16), (7, 5, 100, 64)):
        pc = O.PhaseCfg("exp", tau=T * 0.02 * 0.8, alpha_phase=3.0)
        bc = O.BasisCfg("prodmp", num_basis=nb, alpha=10)
        tc = O.TrajCfg("prodmp", action_dim=D, **flags)
        eng = make_engine(pc, bc, tc, 0.02, T * 0.02)
        params, ip, iv = inputs(pc, bc, tc, B, seed=D)
        # a 32 MB allocation of its own (beyond the caching allocator's pooled sizes); the parameters are its LAST bytes, so
        # one float past them is the first byte of whatever follows -- usually nothing
        slab = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
        n = params.size * 4
        pdev = slab[slab.numel() - n:].view(torch.float32).view(params.shape)
        pdev.copy_(torch.from_numpy(params))
        for opts in ({{}}, {{"mapping": 2}}, {{"mapping": 2, "bulk": 2}}, {{"flat": 1}}, {{"mapping": 1, "ipw": 1}}):
            _lib.reset_options()
            for k, v in opts.items():
                _lib.set_option(k, v)
            pos, vel = eng.trajectory(pdev, ip, iv, 0.0)
            torch.cuda.synchronize()
            rp, _ = O.get_trajectory(pc, bc, tc, params, T * 0.02, 0.02, 0.0, ip, iv, dtype=np.float64)
            assert np.abs(pos.cpu().numpy() - rp).max() <= 2e-5 * np.abs(rp).max(), (flags, D, opts)
        del slab, pdev
print("tail ok")
"""


def test_disabled_goal_or_weights_never_read_past_the_parameter_buffer():
    """
    Round 3's fuzz soak (2 x 10^5 random configurations in one process) died with a GPU memory fault once per ~10^4 cases:
    with disable_goal / disable_weights the B-fragment gather loaded -- unconditionally, the value was never used -- the
    slot where the disabled block's parameters would have been, i.e. up to nb - 1 floats past the last DoF of the last
    episode, past the end of `params`; a fault when the buffer ends on the last mapped page.  The parameters here ARE the
    last bytes of a 32 MB allocation of their own; run in a child process (a fault ends the process, not the suite).
    """
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TAIL_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600,
                       cwd=root)
    assert r.returncode == 0 and "tail ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


_TWO_BIG_LDS_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import dataclasses
import numpy as np, torch
from oracle import mp_oracle as O
from tests.test_gpu_trajectory import CFG2, inputs, make_engine
from fancy_gym_amd import RolloutSpec, _lib
import bench
# two DIFFERENT instantiations with the SAME parameter list, each beyond 64 KB of dynamic LDS, first use of either in this process:
# k_traj_flat<prodmp, act> with 8 contraction columns (cfg2) and with 4 (one basis function + goal + two boundary columns), then the
# trajectory-only pair of k_traj_ring (134 KB) -- round 3 kept the "attribute already raised" flag per function TYPE, so the second
# kernel of a signature never got its own hipFuncSetAttribute call
pc, bc, tc, dt, dur = CFG2
small = dataclasses.replace(bc, num_basis=1)
_lib.set_option("flat", 1)
spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
kernels = []
for b in (bc, small, bc, small):
    eng = make_engine(pc, b, tc, dt, dur)
    params, ip, iv = inputs(pc, b, tc, 300, seed=4)
    cp, cv = np.zeros((300, 7)), np.zeros((300, 7))
    pos, vel, act = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
    torch.cuda.synchronize()
    kernels.append(eng.last_kernel())
    rp, rv = O.get_trajectory(pc, b, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    assert np.abs(pos.cpu().numpy() - rp).max() <= 2e-5 * np.abs(rp).max(), eng.last_kernel()
assert all(k == "k_traj_flat<prodmp,act>" for k in kernels), kernels
_lib.reset_options(); _lib.set_option("ring", 1)
for b in (small, bc):
    eng = make_engine(pc, b, tc, dt, dur)
    params, ip, iv = inputs(pc, b, tc, 300, seed=5)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "k_traj_ring<prodmp>", eng.last_kernel()
    rp, rv = O.get_trajectory(pc, b, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    assert np.abs(pos.cpu().numpy() - rp).max() <= 2e-5 * np.abs(rp).max()
print("two big-LDS kernels ok")
"""


def test_two_kernels_of_one_signature_that_each_need_more_than_64_kb_of_lds_in_one_fresh_process():
    """the once-per-kernel flag of hipFuncSetAttribute(MaxDynamicSharedMemorySize) is keyed on the function
    VALUE and the device; a fresh process launches two instantiations of one signature, both above 64 KB, in both orders of use"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TWO_BIG_LDS_CHILD, root], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "two big-LDS kernels ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.timing
@pytest.mark.parametrize("B", [4096, 16384, 65536, 262144])
def test_the_automatic_kernel_choice_is_within_ten_percent_of_the_best_forced_variant(B):
    """The launcher picks a kernel family by ~25 byte thresholds fitted to THIS chip's caches (mpk_dev.h, mpk_traj_launch.hip); a
    driver or clock change that moves a crossover should fail a test, not wait for the next sweep.  The headline launch (cfg2,
    trajectory + actions) and the closed-loop step at four sizes: every family the launcher could have taken is forced in turn
    (captured graphs of the same launch, warmed by GPU-busy time, timed in alternating rounds: tools/closed_bench.py), and the
    automatic choice must be within 10 % of the fastest."""
    from tools.closed_bench import CFG2 as KW, DG, PG, capture, time_rows
    eng = TrajectoryEngine(device=0, **KW)
    g = torch.Generator().manual_seed(0)
    params = torch.randn((B, eng.num_params), generator=g).cuda()
    ip = (torch.rand((B, 7), generator=g) * 2 - 1).cuda()
    iv = torch.zeros((B, 7), device="cuda")
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(3))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    q, qd = cp.clone(), cv.clone()
    spec_s = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="static")
    spec_d = RolloutSpec("motor", 7, PG, DG, -1.0, 1.0, plant="double_integrator", dt=0.02)
    reps = 20 if B <= 65536 else 4
    cases = {
        "open loop": (lambda: eng.trajectory_actions(params, ip, iv, spec_s, cp, cv, out=out),
                      [{}, {"mapping": 1}, {"mapping": 2}, {"flat": 1}, {"ring": 1}]),
        "closed loop": (lambda: eng.trajectory_rollout(params, ip, iv, spec_d, q, qd, out=out),
                        [{}, {"pipe": 1}, {"quad": 2}, {"quad": 3}, {"ring": 1}]),
    }
    try:
        for name, (fn, variants) in cases.items():
            graphs, kernels = [], []
            for opts in variants:
                _lib.reset_options()
                for k, v in opts.items():
                    _lib.set_option(k, v)
                graphs.append(capture(fn, reps))
                kernels.append(eng.last_kernel())
            _lib.reset_options()
            ts = time_rows(graphs, reps=reps, rounds=7, busy_ms=40.0)
            best = min(range(len(ts)), key=lambda i: ts[i])
            table = ", ".join(f"{opts or 'auto'} {k} {t * 1e6:.1f} us" for opts, k, t in zip(variants, kernels, ts))
            assert ts[0] <= 1.10 * ts[best], f"{name} at B = {B}: automatic choice {kernels[0]} is {ts[0] / ts[best]:.2f} x the best ({table})"
            del graphs
            eng.unpin_tables()
    finally:
        _lib.reset_options()


def test_ticket_counters_of_dead_captures_are_recycled(mpk_option):
    """every (capture id, stream) pair takes one of 4 096 ticket slots of the handle for good, so an application that
    re-captures its step every iteration ran out and fell to static batches silently.  mpk_unpin_tables (the caller's statement that the
    graphs are dead) returns the captures' slots to the pool: 5 000 capture / destroy / unpin rounds of a ring launch keep drawing
    tickets (a launch without a ticket counter is reported by the kernel's name suffix only in debug builds, so the pool itself is
    checked: after the rounds an eager ring launch still equals the reference and a fresh capture replays correctly)"""
    pc, bc, tc, dt, dur = cfg_for("prodmp", 7, 5, 100)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = 12288
    params, ip, iv = inputs(pc, bc, tc, B, seed=6)
    P, IP, IV = (torch.tensor(x, device="cuda") for x in (params, ip, iv))
    out = tuple(torch.empty((B, 100, 7), device="cuda") for _ in range(2))
    ref = [x.clone() for x in eng.trajectory(P, IP, IV, 0.0)]
    mpk_option("ring", 1)
    eng.trajectory(P, IP, IV, 0.0, out=out)
    assert eng.last_kernel().startswith("k_traj_ring")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for it in range(5000):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                eng.trajectory(P, IP, IV, 0.0, out=out)
        if it % 1000 == 0:
            out[0].zero_(); out[1].zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), it
        del g
        eng.unpin_tables()
    out[0].zero_(); out[1].zero_()
    eng.trajectory(P, IP, IV, 0.0, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
