"""
GPU parity tests proper: the HIP path (through the C-ABI, via fancy_gym_amd.TrajectoryEngine) against the CPU oracle on
the same seeded inputs.  Tolerance: BASELINE.json north_star -- 1e-5 relative in fp32; stated as
|gpu - oracle| <= 1e-5 * max|oracle| + 1e-5 * |oracle| per output array.  Integer parts (table indices) bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import mp_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, ref, name, rtol=RTOL, atol=0.0):
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = np.abs(ref).max() if ref.size else 1.0
    err = np.abs(got - ref)
    tol = rtol * scale + rtol * np.abs(ref) + atol
    bad = err > tol
    assert not bad.any(), f"{name}: max err {err.max():.3e} (scale {scale:.3e}), {bad.sum()} / {bad.size} outside tol"


def fd_atol(pos_ref, dt):
    """
    ProMP velocity is a forward difference of fp32 positions: its conditioning is 1/dt, so two
    formulations that agree on every position to <= 1 ulp may differ in velocity by 2 ulp(max|pos|) / dt.
    That term is added to the 1e-5 relative tolerance for ProMP velocities only.
    """
    return 2.0 * float(np.finfo(np.float32).eps) * float(np.abs(pos_ref).max()) / dt


def make_engine(pc, bc, tc, dt, duration, **kw):
    from fancy_gym_amd import TrajectoryEngine
    return TrajectoryEngine(
        tc.trajectory_generator_type, pc.phase_generator_type, bc.basis_generator_type, tc.action_dim, bc.num_basis,
        dt=dt, duration=duration, tau=pc.tau, delay=pc.delay, alpha_phase=pc.alpha_phase, learn_tau=pc.learn_tau,
        learn_delay=pc.learn_delay, tau_bound=pc.tau_bound, delay_bound=pc.delay_bound,
        basis_bandwidth_factor=bc.basis_bandwidth_factor, num_basis_outside=bc.num_basis_outside,
        num_basis_zero_start=bc.num_basis_zero_start if bc.basis_generator_type == "zero_rbf" else 0,
        num_basis_zero_goal=bc.num_basis_zero_goal if bc.basis_generator_type == "zero_rbf" else 0,
        basis_alpha=bc.alpha, basis_dt=bc.dt, pre_compute_length_factor=bc.pre_compute_length_factor,
        weights_scale=tc.weights_scale, goal_scale=tc.goal_scale, dmp_alpha=tc.alpha,
        auto_scale_basis=tc.auto_scale_basis, relative_goal=tc.relative_goal, disable_goal=tc.disable_goal,
        disable_weights=tc.disable_weights, relative_goal_mode=tc.relative_goal_mode,
        goal_offset_mode=tc.goal_offset_mode, goal_offset=tc.goal_offset, dmp_first_sample=tc.dmp_first_sample,
        single_rbf_mode=bc.single_rbf_mode, **kw)


def inputs(pc, bc, tc, B, seed=0):
    rng = np.random.default_rng(seed)
    P = O.num_params(pc, bc, tc)
    params = rng.standard_normal((B, P)).astype(np.float32)
    i = 0
    if pc.learn_tau:
        lo, hi = pc.tau_bound
        params[:, i] = rng.uniform(lo, min(hi, lo + 2.0), B); i += 1
    if pc.learn_delay:
        lo, hi = pc.delay_bound
        params[:, i] = rng.uniform(lo, min(hi, lo + 0.5), B); i += 1
    ip = rng.uniform(-1, 1, (B, tc.action_dim)).astype(np.float32)
    iv = rng.uniform(-1, 1, (B, tc.action_dim)).astype(np.float32)
    return params, ip, iv


CFG2 = (O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0),
        O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10),
        O.TrajCfg("prodmp", action_dim=7), 0.02, 2.0)
CFG3 = (O.PhaseCfg("exp", tau=4.0, alpha_phase=2.0),
        O.BasisCfg("rbf", num_basis=5, basis_bandwidth_factor=3),
        O.TrajCfg("dmp", action_dim=7, alpha=25.0), 0.02, 4.0)
CFG1 = (O.PhaseCfg("linear", tau=4.0),
        O.BasisCfg("zero_rbf", num_basis=5, num_basis_zero_start=1, num_basis_zero_goal=0, basis_bandwidth_factor=3),
        O.TrajCfg("promp", action_dim=5), 0.02, 4.0)
CFG5 = (O.PhaseCfg("linear", tau=2.8),
        O.BasisCfg("zero_rbf", num_basis=3, num_basis_zero_start=1, num_basis_zero_goal=1, basis_bandwidth_factor=3),
        O.TrajCfg("promp", action_dim=7), 0.008, 2.8)
CFG4 = (O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0),
        O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=3, alpha=10),
        O.TrajCfg("prodmp", action_dim=7, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True, disable_goal=True),
        0.02, 2.0)
SHARED = {"cfg1_promp_reacher5d": CFG1, "cfg2_prodmp_boxpushing": CFG2, "cfg3_dmp_reacher7d": CFG3,
          "cfg4_prodmp_replan": CFG4, "cfg5_promp_tabletennis": CFG5}


@pytest.mark.parametrize("name", list(SHARED))
@pytest.mark.parametrize("B", [1, 3, 64, 1000])
@pytest.mark.parametrize("init_time", [0.0, 0.5])
def test_shared_phase_matches_oracle(name, B, init_time):
    pc, bc, tc, dt, duration = SHARED[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("k_traj_")
    for dtype in (np.float64, np.float32):
        rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, init_time, ip, iv, dtype=dtype)
        close(pos.cpu().numpy(), rp, f"{name} pos vs oracle {dtype.__name__}")
        close(vel.cpu().numpy(), rv, f"{name} vel vs oracle {dtype.__name__}",
              atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)


# ---- DMP with a shared phase: the response route (round 5) and the serial explicit-Euler kernels behind "dmp_response" 0 ---------
DMP_SHAPES = {
    "cfg3": CFG3,
    "dmp_first_is_step": (O.PhaseCfg("exp", tau=4.0, alpha_phase=2.0), O.BasisCfg("rbf", num_basis=5),
                          O.TrajCfg("dmp", action_dim=7, alpha=25.0, dmp_first_sample="step"), 0.02, 4.0),
    "dmp_scaled_delay": (O.PhaseCfg("exp", tau=1.0, delay=0.2, alpha_phase=3.0), O.BasisCfg("rbf", num_basis=10),
                         O.TrajCfg("dmp", action_dim=3, alpha=20.0, weights_scale=0.7, goal_scale=1.3), 0.02, 1.4),
    "dmp_13_columns": (O.PhaseCfg("exp", tau=2.0), O.BasisCfg("rbf", num_basis=13),
                       O.TrajCfg("dmp", action_dim=16, alpha=25.0), 0.01, 1.0),
}


@pytest.mark.parametrize("name", list(DMP_SHAPES))
@pytest.mark.parametrize("B", [1, 5, 1000])
@pytest.mark.parametrize("init_time", [0.0, 0.5])
def test_dmp_response_route_and_serial_kernels_match_the_oracle_and_each_other(name, B, init_time, mpk_option):
    """The explicit Euler recurrence is linear in (w, g, y_b, v_b): with a shared phase the library contracts the parameters with the
    response rows of THE SAME Euler map (k_build_shared, float64, rounded once) on the matrix-core kernels; "dmp_response" 0 keeps the
    serial kernels that run the reference's fp32 recurrence operation for operation.  Both against the float64 and the float32
    oracle at the 1e-5 contract, and against each other at 2e-6 of the scale."""
    pc, bc, tc, dt, duration = DMP_SHAPES[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + 11)
    iv = (np.random.default_rng(B).uniform(-0.5, 0.5, iv.shape)).astype(np.float32)
    outs = {}
    for resp in (1, 0):
        mpk_option("dmp_response", -1 if resp else 0)
        pos, vel = eng.trajectory(params, ip, iv, init_time)
        torch.cuda.synchronize()
        assert ("dmp_resp" in eng.last_kernel()) == bool(resp), eng.last_kernel()
        outs[resp] = (pos.cpu().numpy(), vel.cpu().numpy())
        for dtype in (np.float64, np.float32):
            rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, init_time, ip, iv, dtype=dtype)
            close(outs[resp][0], rp, f"{name} pos vs oracle {dtype.__name__} (response {resp})")
            close(outs[resp][1], rv, f"{name} vel vs oracle {dtype.__name__} (response {resp})")
    for a, b in zip(outs[1], outs[0]):
        assert np.abs(a.astype(np.float64) - b).max() <= 2e-6 * np.abs(b).max()


def test_dmp_response_route_is_left_where_the_euler_map_is_not_comfortably_stable():
    """alpha ds > 1 (here 25 x 0.02 / 0.3 = 1.67: past the explicit Euler map's stability bound, the reference's own recurrence rings
    or diverges): the serial kernels, which reproduce that recurrence operation for operation -- checked against the float32 oracle"""
    pc, bc, tc = O.PhaseCfg("exp", tau=0.3), O.BasisCfg("rbf", num_basis=5), O.TrajCfg("dmp", action_dim=4, alpha=25.0)
    eng = make_engine(pc, bc, tc, 0.02, 0.4)
    params, ip, iv = inputs(pc, bc, tc, 33, seed=3)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    assert "dmp_resp" not in eng.last_kernel() and "dmp" in eng.last_kernel(), eng.last_kernel()
    rp, rv = O.get_trajectory(pc, bc, tc, params, 0.4, 0.02, 0.0, ip, iv, dtype=np.float32)
    close(pos.cpu().numpy(), rp, "pos"); close(vel.cpu().numpy(), rv, "vel")


@pytest.mark.parametrize("init_time", [0.0, 0.02, 0.5, 1.0, 1.5])
def test_prodmp_indices_bit_exact(init_time):
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    idx, idxb = eng.prodmp_indices(init_time)
    tabs = O.prodmp_tables(pc, bc, np.float32)
    times = O.make_times(duration, dt, init_time, dtype=np.float32)
    ref = O.prodmp_indices(times, np.float32(pc.tau), np.float32(pc.delay), tabs.scaled_dt)
    refb = O.prodmp_indices(np.array([init_time], np.float32), np.float32(pc.tau), np.float32(pc.delay), tabs.scaled_dt)
    assert np.array_equal(idx, ref.astype(np.int32))
    assert idxb == int(refb[0])
    assert np.array_equal(eng.times(), O.make_times(duration, dt, 0.0, dtype=np.float32))


def test_prodmp_tables_match_oracle_f64():
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    t = eng.prodmp_tables()
    ref = O.prodmp_tables(pc, bc, np.float64)
    for k, r in (("y1", ref.y1), ("y2", ref.y2), ("dy1", ref.dy1), ("dy2", ref.dy2), ("pos_basis", ref.pos_basis),
                 ("vel_basis", ref.vel_basis), ("scale", ref.scale_factors)):
        np.testing.assert_allclose(t[k], r, rtol=1e-11, atol=1e-12, err_msg=k)


PER_ROW = {
    "prodmp_learn_tau_delay": (O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True,
                                          tau_bound=(0.8, 1.5), delay_bound=(0.05, 0.15)),
                               O.BasisCfg("prodmp", num_basis=3, basis_bandwidth_factor=3, alpha=25),
                               O.TrajCfg("prodmp", action_dim=7, weights_scale=0.7, auto_scale_basis=True,
                                         relative_goal=True, disable_goal=True), 0.008, 2.8),
    "promp_learn_tau": (O.PhaseCfg("linear", tau=1.0, learn_tau=True, tau_bound=(0.04, 1.0)),
                        O.BasisCfg("rbf", num_basis=10), O.TrajCfg("promp", action_dim=2), 0.02, 1.0),
    "dmp_learn_delay": (O.PhaseCfg("exp", tau=1.0, learn_delay=True, delay_bound=(0.0, 0.96)),
                        O.BasisCfg("rbf", num_basis=10), O.TrajCfg("dmp", action_dim=3), 0.02, 1.0),
}


@pytest.mark.parametrize("name", list(PER_ROW))
@pytest.mark.parametrize("B", [1, 5, 257])
def test_per_episode_phase_matches_oracle(name, B):
    pc, bc, tc, dt, duration = PER_ROW[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + 1)
    pos, vel = eng.trajectory(params, ip, iv, 0.0)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("k_traj_phase")
    rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, 0.0, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, f"{name} pos")
    close(vel.cpu().numpy(), rv, f"{name} vel", atol=fd_atol(rp, dt) if "promp" in name else 0.0)


@pytest.mark.parametrize("name", ["cfg2", "cfg5", "cfg3"])
def test_per_episode_init_time_equals_shared_path(name):
    """promp / dmp: the per-episode kernel builds its rows with the functions of the shared-table builder and runs the
    fmaf chain in the MFMA's accumulation order -> identical bits.  prodmp: the per-episode kernel evaluates the
    reference's c1*y1 + c2*y2 + Psi.wg form, the shared-phase kernels the folded one -> equal to rounding"""
    pc, bc, tc, dt, duration = {"cfg2": CFG2, "cfg5": CFG5, "cfg3": CFG3}[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    B = 130
    params, ip, iv = inputs(pc, bc, tc, B, seed=7)
    t0 = 0.5 if name == "cfg2" else 0.0
    p0, v0 = eng.trajectory(params, ip, iv, t0)
    it = torch.full((B,), t0, dtype=torch.float32, device="cuda")
    p1, v1 = eng.trajectory(params, ip, iv, it)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("k_traj_phase")
    if name in ("cfg2", "cfg3"):
        # (cfg3: the shared phase takes the response route since round 5 -- bit identity holds WITHIN the shared-phase families and
        # within the per-episode families, the 1e-5 contract across them; with "dmp_response" 0 the serial kernels are bit-identical)
        for a, b in ((p0, p1), (v0, v1)):
            a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
            assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
    else:
        assert torch.equal(p0, p1)
        assert torch.equal(v0, v1)

# ---- k_traj_flat: whole-trajectory LDS images, flat stores (the HBM-streaming open-loop kernel of round 3) ----------------
FLAT_PROMP = (O.PhaseCfg("linear", tau=2.8),
              O.BasisCfg("zero_rbf", num_basis=3, num_basis_zero_start=1, num_basis_zero_goal=1, basis_bandwidth_factor=3),
              O.TrajCfg("promp", action_dim=4), 0.028, 2.8)          # T * D = 400: whole float4 chunks, four episodes per group
FLAT_PROMP5 = (O.PhaseCfg("linear", tau=2.0),
               O.BasisCfg("zero_rbf", num_basis=5, num_basis_zero_start=1, num_basis_zero_goal=0, basis_bandwidth_factor=3),
               O.TrajCfg("promp", action_dim=5), 0.02, 2.0)          # cfg1's basis on a 100-step horizon


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "promp4", "promp5"])
@pytest.mark.parametrize("B", [1, 2, 3, 64, 1001])
@pytest.mark.parametrize("init_time", [0.0, 0.5])
def test_flat_kernel_matches_the_oracle_and_the_other_kernels_bitwise(name, B, init_time, mpk_option):
    """same tile arithmetic as k_traj_stream / k_traj_tiles (same device functions) -> identical bits; and the oracle"""
    pc, bc, tc, dt, duration = {"cfg2": CFG2, "cfg4": CFG4, "promp4": FLAT_PROMP, "promp5": FLAT_PROMP5}[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + 3)
    mpk_option("flat", 1)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("k_traj_flat"), eng.last_kernel()
    rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, init_time, ip, iv, dtype=np.float64)
    close(pos.cpu().numpy(), rp, f"{name} pos")
    close(vel.cpu().numpy(), rv, f"{name} vel", atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
    for mapping in (1, 2):
        mpk_option("flat", 0); mpk_option("mapping", mapping)
        p2, v2 = eng.trajectory(params, ip, iv, init_time)
        assert not eng.last_kernel().startswith("k_traj_flat")
        assert torch.equal(pos, p2) and torch.equal(vel, v2), (mapping, eng.last_kernel())


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "promp4", "promp5"])
@pytest.mark.parametrize("B", [1, 3, 130, 2051])
def test_flat_kernel_with_the_dof_count_compiled_in_equals_the_generic_one(name, B, mpk_option):
    """5 and 7 DoF with <= 8 contraction columns run k_traj_flat_d (the ring's contraction with immediates + its chunked flush);
    "ring_dbg" 64 selects the generic kernel, 32 the generic (interleaved) flush: the same bits, trajectory and fused actions"""
    import bench
    from fancy_gym_amd import RolloutSpec
    pc, bc, tc, dt, duration = {"cfg2": CFG2, "cfg4": CFG4, "promp4": FLAT_PROMP, "promp5": FLAT_PROMP5}[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    D = tc.action_dim
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + 11)
    rng = np.random.default_rng(B)
    cp, cv = rng.uniform(-1, 1, (B, D)), rng.uniform(-1, 1, (B, D))
    spec = RolloutSpec("motor", D, bench.P_GAINS[:D], bench.D_GAINS[:D], -1.0, 1.0, plant="static")
    mpk_option("flat", 1)
    outs = []
    for dbg in (64, 0, 32):
        mpk_option("ring_dbg", dbg)
        t = [x.clone() for x in eng.trajectory(params, ip, iv, 0.25)]
        assert eng.last_kernel().startswith("k_traj_flat"), eng.last_kernel()
        a = [x.clone() for x in eng.trajectory_actions(params, ip, iv, spec, cp, cv, init_time=0.25)]
        # (three whole-trajectory images of promp4's 350-step episodes exceed the flat kernels' LDS budget: k_traj_stream there)
        assert eng.last_kernel().startswith("k_traj_flat" if name != "promp4" else "k_traj_stream"), eng.last_kernel()
        outs.append(t + a)
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))


@pytest.mark.parametrize("B", [1, 5, 130, 4096])
@pytest.mark.parametrize("ctrl", ["motor", "position", "velocity"])
def test_flat_kernel_fused_actions_bit_exact(B, ctrl, mpk_option):
    """trajectory + tracking-controller actions (frozen state) through k_traj_flat: actions bit-exact against the oracle's
    float64 controller and against the tile-major kernel"""
    import bench
    from fancy_gym_amd import RolloutSpec
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    rng = np.random.default_rng(B)
    cp, cv = rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))
    spec = RolloutSpec(ctrl, 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
    mpk_option("flat", 1)
    pos, vel, act = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
    torch.cuda.synchronize()
    assert eng.last_kernel() == "k_traj_flat<prodmp,act>", eng.last_kernel()
    ra, _, _ = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), ctrl, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, "static", dt,
                         cp, cv)
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
    mpk_option("flat", 0)
    p2, v2, a2 = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
    assert eng.last_kernel().startswith("k_traj_tiles") or eng.last_kernel().startswith("k_traj_stream")
    assert torch.equal(pos, p2) and torch.equal(vel, v2) and torch.equal(act, a2)


# ---- k_traj_ring: producer waves + store-engine waves, in-order batch tickets (round 4) --------------------------------------------
RING_OPTS = [dict(), dict(ring_np=4, ring_ns=4), dict(ring_np=10, ring_ns=1, ring_m=2), dict(ring_m=1, ring_parts=3),
             dict(ring_dbg=16), dict(ring_dbg=4), dict(ring_dbg=32), dict(ring_np=3, ring_ns=3, ring_m=3, ring_parts=2)]


@pytest.mark.parametrize("name", ["cfg2", "cfg4", "cfg5", "cfg1", "promp4", "promp5"])
@pytest.mark.parametrize("B", [1, 2, 7, 64, 1001, 4107])
def test_ring_kernel_matches_the_other_kernels_bitwise_and_the_oracle(name, B, mpk_option):
    """k_traj_ring under every launch geometry (producers / engine waves / groups per batch / waves per group), its three batch
    orders (tickets from one counter, static ranges, b % gridDim) and both contraction loops (compile-time DoF count with
    immediates, generic): the bits of k_traj_tiles / k_traj_stream.  cfg5 (350 x 7: T * D = 2 mod 4, an image of one group fills a
    batch buffer, ragged last batch leaves two floats) and cfg1 (200 x 5) are the shapes k_traj_flat cannot take."""
    pc, bc, tc, dt, duration = {"cfg2": CFG2, "cfg4": CFG4, "cfg5": CFG5, "cfg1": CFG1, "promp4": FLAT_PROMP,
                                "promp5": FLAT_PROMP5}[name]
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B + 11)
    init_time = 0.5 if name in ("cfg2", "cfg4") else 0.0
    mpk_option("flat", 0); mpk_option("ring", 0); mpk_option("mapping", 2)
    p0, v0 = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("k_traj_stream"), eng.last_kernel()
    rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, init_time, ip, iv, dtype=np.float64)
    close(p0.cpu().numpy(), rp, f"{name} pos")
    mpk_option("mapping", -1)
    for opts in RING_OPTS:
        for k in ("ring_np", "ring_ns", "ring_m", "ring_parts", "ring_dbg"):
            mpk_option(k, opts.get(k, -1))
        mpk_option("ring", 1)
        out = (torch.full_like(p0, float("nan")), torch.full_like(v0, float("nan")))
        eng.trajectory(params, ip, iv, init_time, out=out)
        torch.cuda.synchronize()
        assert eng.last_kernel().startswith("k_traj_ring"), (opts, eng.last_kernel())
        assert torch.equal(out[0].view(torch.int32), p0.view(torch.int32)), (name, B, opts)
        assert torch.equal(out[1].view(torch.int32), v0.view(torch.int32)), (name, B, opts)


@pytest.mark.parametrize("B", [1, 5, 130, 4099])
@pytest.mark.parametrize("ctrl", ["motor", "position", "velocity"])
def test_ring_kernel_fused_actions_bit_exact(B, ctrl, mpk_option):
    """trajectory + tracking-controller actions through k_traj_ring (cfg2 and cfg5 shapes): actions bit-exact against the oracle's
    float64 controller, everything bit-identical to the other kernels"""
    import bench
    from fancy_gym_amd import RolloutSpec
    for cfg in (CFG2, CFG5):
        pc, bc, tc, dt, duration = cfg
        eng = make_engine(pc, bc, tc, dt, duration)
        params, ip, iv = inputs(pc, bc, tc, B, seed=B)
        rng = np.random.default_rng(B)
        cp, cv = rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))
        spec = RolloutSpec(ctrl, 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
        mpk_option("ring", 0); mpk_option("flat", 0)
        p2, v2, a2 = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
        assert eng.last_kernel().startswith("k_traj_tiles") or eng.last_kernel().startswith("k_traj_stream")
        ra, _, _ = O.rollout(p2.cpu().numpy(), v2.cpu().numpy(), ctrl, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, "static", dt, cp, cv)
        assert np.array_equal(a2.cpu().numpy(), ra.astype(np.float32))
        for opts in (dict(), dict(ring_dbg=32), dict(ring_np=5, ring_ns=3, ring_m=2), dict(ring_dbg=16, ring_ns=1)):
            for k in ("ring_np", "ring_ns", "ring_m", "ring_parts", "ring_dbg"):
                mpk_option(k, opts.get(k, -1))
            mpk_option("ring", 1)
            pos, vel, act = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
            torch.cuda.synchronize()
            assert eng.last_kernel().startswith("k_traj_ring") and "act" in eng.last_kernel(), eng.last_kernel()
            assert torch.equal(pos, p2) and torch.equal(vel, v2) and torch.equal(act, a2), (opts, B, ctrl)


def test_ring_kernel_is_the_automatic_choice_for_launches_that_stream_to_hbm_and_only_those(mpk_option):
    """outputs beyond kRingBytes (346 MB) of an open-loop promp / prodmp launch WITH actions -> k_traj_ring; trajectory-only launches of
    the shapes k_traj_flat takes stay on it up to kRingTrajBytes (round 5: two workgroups of whole-trajectory images per CU beat the ring
    up to ~3 GB); DMP, the closed loop and smaller launches keep their kernels; two launches of one handle on two streams do not share
    a ticket counter"""
    import bench
    from fancy_gym_amd import RolloutSpec
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    B = 120000                                               # 2 x 336 MB of (pos, vel)
    params, ip, iv = inputs(pc, bc, tc, B, seed=2)
    p2, v2 = [x.clone() for x in eng.trajectory(params, ip, iv, 0.0)]
    assert eng.last_kernel() == "k_traj_flat<prodmp>", eng.last_kernel()
    rng = np.random.default_rng(5)
    cp, cv = rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))
    spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
    pos, vel, act = [x.clone() for x in eng.trajectory_actions(params, ip, iv, spec, cp, cv)]
    assert eng.last_kernel() == "k_traj_ring<prodmp,act>", eng.last_kernel()
    assert torch.equal(pos, p2) and torch.equal(vel, v2)
    mpk_option("ring", 0)
    p1, v1, a1 = eng.trajectory_actions(params, ip, iv, spec, cp, cv)
    assert eng.last_kernel().startswith("k_traj_flat"), eng.last_kernel()
    assert torch.equal(pos, p1) and torch.equal(vel, v1) and torch.equal(act, a1)
    mpk_option("ring", 1)
    p1, v1 = eng.trajectory(params, ip, iv, 0.0)
    assert eng.last_kernel() == "k_traj_ring<prodmp>", eng.last_kernel()
    assert torch.equal(p1, p2) and torch.equal(v1, v2)
    del p1, v1, a1
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for st in (s1, s2, s1, s2):
        with torch.cuda.stream(st):
            outs.append(eng.trajectory(params, ip, iv, 0.0))      # (the ring, forced: four launches, two streams)
    torch.cuda.synchronize()
    for p3, v3 in outs:
        assert torch.equal(p3, p2) and torch.equal(v3, v2)
    mpk_option("ring", -1)
    small = eng.trajectory(params[:4096], ip[:4096], iv[:4096], 0.0)
    assert eng.last_kernel().startswith("k_traj_tiles")
    assert torch.equal(small[0], p2[:4096])
    pc, bc, tc, dt, duration = CFG3
    eng3 = make_engine(pc, bc, tc, dt, duration)
    prm3, ip3, iv3 = inputs(pc, bc, tc, 60000, seed=2)
    eng3.trajectory(prm3, ip3, iv3, 0.0)                      # 60 000 episodes of cfg3: 672 MB -> the ring, on the response route
    assert eng3.last_kernel() == "k_traj_ring<dmp_resp>", eng3.last_kernel()
    mpk_option("dmp_response", 0)
    eng3.trajectory(prm3, ip3, iv3, 0.0)
    assert "<dmp>" in eng3.last_kernel() and "ring" not in eng3.last_kernel(), eng3.last_kernel()


def test_flat_kernel_is_skipped_where_it_does_not_apply(mpk_option):
    """T * D not a multiple of 4 (cfg5: 350 x 7), DMP, images beyond the LDS budget (cfg1: two 200-step 5-DoF episodes per
    wave and array): the forced option falls through to the other kernels"""
    mpk_option("flat", 1)
    for cfg in (CFG5, CFG3, CFG1):
        pc, bc, tc, dt, duration = cfg
        eng = make_engine(pc, bc, tc, dt, duration)
        params, ip, iv = inputs(pc, bc, tc, 9, seed=1)
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        assert not eng.last_kernel().startswith("k_traj_flat")
        rp, rv = O.get_trajectory(pc, bc, tc, params, duration, dt, 0.0, ip, iv, dtype=np.float64)
        close(pos.cpu().numpy(), rp, "pos")


@pytest.mark.parametrize("B", [700, 12288, 70001])
def test_ring_tickets_of_every_size_leave_the_same_bits(B, mpk_option):
    """Batches per ticket of the ring's device counter ("ring_tb"; automatic: >= 192 KB of output per ticket but at least ~16 tickets per
    workgroup).  Round 5 found wave 0 waiting for a ticket it was itself about to publish when a ticket covered less than a round of
    producers (tickets of one batch: the launch timed out -- and said so); the kernel keeps ceil(2 NP / units-per-ticket) + 2 tickets
    in flight now and the launcher never hands it tickets finer than NP / 2 units.  Every ticket size, with one to four groups per
    batch and three to twelve producers: the bits of the tile-major kernel, and no fault."""
    pc, bc, tc, dt, duration = CFG2
    eng = make_engine(pc, bc, tc, dt, duration)
    params, ip, iv = inputs(pc, bc, tc, B, seed=B)
    mpk_option("ring", 0); mpk_option("flat", 0); mpk_option("mapping", 1)
    p0, v0 = [x.clone() for x in eng.trajectory(params, ip, iv, 0.0)]
    assert eng.last_kernel().startswith("k_traj_tiles")
    mpk_option("mapping", -1); mpk_option("flat", -1); mpk_option("ring", 1)
    for tb in (-1, 1, 2, 5, 64):
        for m, np_ in ((-1, -1), (1, 8), (2, 12), (4, 3), (1, 14)):
            mpk_option("ring_tb", tb); mpk_option("ring_m", m); mpk_option("ring_np", np_)
            pos, vel = eng.trajectory(params, ip, iv, 0.0)
            assert eng.last_kernel() == "k_traj_ring<prodmp>", eng.last_kernel()
            assert torch.equal(pos, p0) and torch.equal(vel, v0), (tb, m, np_)
    eng.check_range()          # (synchronises; a ring time-out of any launch above would be reported here)
