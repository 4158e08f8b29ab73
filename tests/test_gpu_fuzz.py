"""
GPU suite, part 4: seeded random configurations (movement primitive, DoF, basis count, horizon, dt, phase constants,
ProDMP flags, batch size, init_time, work decomposition) against the float64 oracle -- every kernel variant gets hit
with shapes nobody hand-picked.
"""
import dataclasses

import numpy as np
import pytest
import torch

from fancy_gym_amd import RolloutSpec
from oracle import mp_oracle as O
from tests.test_gpu_trajectory import close, fd_atol, inputs, make_engine

pytestmark = pytest.mark.gpu


def random_case(rng):
    mp = rng.choice(["prodmp", "promp", "dmp"])
    D = int(rng.integers(1, 17))
    T = int(rng.integers(2, 131))
    dt = float(rng.choice([0.008, 0.01, 0.02, 0.05]))
    dur = T * dt
    learn = bool(rng.random() < 0.25)
    pkw = {}
    if learn:
        pkw = dict(learn_tau=bool(rng.random() < 0.7), learn_delay=bool(rng.random() < 0.5))
        if not (pkw["learn_tau"] or pkw["learn_delay"]):
            pkw["learn_tau"] = True
        pkw.update(tau_bound=(0.4 * dur, dur), delay_bound=(0.0, 0.3 * dur))
    if mp == "prodmp":
        nb = int(rng.integers(1, 14))
        pc = O.PhaseCfg("exp", tau=float(rng.uniform(0.5, 1.0)) * dur, alpha_phase=float(rng.uniform(1.5, 4.0)), **pkw)
        bc = O.BasisCfg("prodmp", num_basis=nb, alpha=float(rng.choice([10.0, 25.0])),
                        basis_bandwidth_factor=float(rng.uniform(1.5, 4.0)))
        dg, dw = bool(rng.random() < 0.3), False
        if not dg:
            dw = bool(rng.random() < 0.15)
        tc = O.TrajCfg("prodmp", action_dim=D, weights_scale=float(rng.uniform(0.2, 1.5)),
                       goal_scale=float(rng.uniform(0.2, 1.5)), auto_scale_basis=bool(rng.random() < 0.4),
                       relative_goal=bool(rng.random() < 0.4), disable_goal=dg, disable_weights=dw)
    elif mp == "promp":
        nb = int(rng.integers(1, 13))
        zero = bool(rng.random() < 0.6)
        pc = O.PhaseCfg(str(rng.choice(["linear", "exp"])), tau=float(rng.uniform(0.5, 1.0)) * dur,
                        alpha_phase=float(rng.uniform(1.5, 4.0)), **pkw)
        bc = O.BasisCfg("zero_rbf" if zero else "rbf", num_basis=nb, num_basis_zero_start=int(rng.integers(0, 3)) if zero else 0,
                        num_basis_zero_goal=int(rng.integers(0, 2)) if zero else 0,
                        num_basis_outside=0 if zero or nb < 4 else int(rng.integers(0, 2)),
                        basis_bandwidth_factor=float(rng.uniform(1.5, 4.0)))
        tc = O.TrajCfg("promp", action_dim=D, weights_scale=float(rng.uniform(0.2, 1.5)))
        T = max(T, 2)
    else:
        nb = int(rng.integers(1, 17))
        pc = O.PhaseCfg("exp", tau=float(rng.uniform(0.7, 1.0)) * dur, alpha_phase=float(rng.uniform(1.5, 3.0)), **pkw)
        bc = O.BasisCfg("rbf", num_basis=nb, basis_bandwidth_factor=float(rng.uniform(2.0, 4.0)))
        tc = O.TrajCfg("dmp", action_dim=D, alpha=float(rng.choice([10.0, 25.0])),
                       weights_scale=float(rng.uniform(0.5, 20.0)), goal_scale=float(rng.uniform(0.5, 1.5)))
    B = int(rng.choice([1, 2, 3, 5, 16, 33, 100, 257]))
    init_time = 0.0 if rng.random() < 0.5 else float(rng.integers(1, max(2, T // 3))) * dt
    return pc, bc, tc, dt, dur, B, init_time


import os

# MPK_FUZZ_CASES=2000 python -m pytest tests/test_gpu_fuzz.py -m gpu   for a longer soak
N_CASES = int(os.environ.get("MPK_FUZZ_CASES", "120"))
START = int(os.environ.get("MPK_FUZZ_START", "0"))


@pytest.mark.parametrize("seed", range(START, START + N_CASES))
def test_random_configuration_matches_oracle(seed, monkeypatch, mpk_option):
    rng = np.random.default_rng(10_000 + seed)
    pc, bc, tc, dt, dur, B, init_time = random_case(rng)
    mpk_option("mapping", str(rng.choice(["0", "1", "2"])))
    mpk_option("bulk", str(rng.choice(["0", "2"])))
    mpk_option("quad", str(rng.choice(["0", "1", "2", "3", "4"])))
    mpk_option("pd_quad", str(rng.choice(["0", "1", "2"])))
    mpk_option("phase_chunk", str(rng.choice(["1", "2", "4"])))
    mpk_option("phase_table", str(rng.choice(["0", "1"])))
    wt = str(rng.choice(["", "0", "1"]))                 # store cache policy: automatic / plain / write-through
    if wt:
        mpk_option("write_through", wt)
    else:
        mpk_option("write_through", -1)
    # the Appendix-A options and the round-2 kernel options, from a generator of their own (cases keep their shapes)
    r2 = np.random.default_rng(77_000 + seed)
    mpk_option("pipe", int(r2.choice([-1, 0, 1])))
    mpk_option("split", int(r2.choice([-1, 0, 1])))
    r3 = np.random.default_rng(99_000 + seed)       # round 3: flat rounds of the per-episode prodmp kernel, chunks up to 8
    mpk_option("phase_flat", int(r3.choice([-1, 0, 1])))
    if r3.integers(0, 2):
        mpk_option("phase_chunk", int(r3.integers(1, 9)))
    r5 = np.random.default_rng(55_000 + seed)       # round 5: DMP with a shared phase on the response route / the serial kernels
    mpk_option("dmp_response", int(r5.choice([-1, -1, 0])))
    r6 = np.random.default_rng(66_000 + seed)       # round 6: k_phase_fused's tile split (frozen-state actions) and producer / consumer form
    mpk_option("phase_split", int(r6.choice([-1, -1, 1, 2, 3, 8])))
    mpk_option("phase_pipe", int(r6.choice([-1, 0, 1])))
    mpk_option("pd_pipe", int(r6.choice([-1, 0, 1])))          # (the rollout on existing trajectories as a producer / consumer workgroup)
    if tc.trajectory_generator_type == "prodmp":
        tc = dataclasses.replace(tc, relative_goal_mode=str(r2.choice(["after_scale", "before_scale"])),
                                 goal_offset_mode=str(r2.choice(["ignore", "add"])), goal_offset=float(r2.uniform(-0.5, 0.5)))
    elif tc.trajectory_generator_type == "dmp":
        tc = dataclasses.replace(tc, dmp_first_sample=str(r2.choice(["init", "step"])))
    if tc.trajectory_generator_type == "prodmp":
        # keep the plan inside the pre-computed range (6 tau): reference raises otherwise
        tau_min = pc.tau_bound[0] if pc.learn_tau else pc.tau
        if (dur + init_time) / tau_min > 5.9:
            init_time = 0.0
        if dur / tau_min > 5.9:
            pytest.skip("beyond the ProDMP pre-computation range")
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    if pc.learn_tau:
        params[:, 0] = rng.uniform(pc.tau_bound[0], pc.tau_bound[1], B)
    if pc.learn_delay:
        params[:, int(pc.learn_tau)] = rng.uniform(pc.delay_bound[0], pc.delay_bound[1], B)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    assert np.isfinite(rp).all()
    # conditioning allowance: where the reference's OWN fp32 arithmetic (the fp32 oracle) is further than 2e-6 of the
    # scale from the exact result -- cancellation among terms much larger than the trajectory, e.g. a ProDMP velocity
    # next to the boundary condition with a tiny tau (case 11003: fp32 reference 1.7e-4 off, GPU 1.7e-5) -- "within
    # 1e-5 of the reference" cannot be resolved finer than that error, so it is added to the tolerance
    p32, v32 = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float32)

    def slack(r32, r64):
        e = float(np.abs(r32.astype(np.float64) - r64).max()) if r64.size else 0.0
        return e if e > 2e-6 * float(np.abs(r64).max()) else 0.0
    close(pos.cpu().numpy(), rp, f"pos [{eng.last_kernel()}]", atol=slack(p32, rp))
    fd = tc.trajectory_generator_type == "promp"
    close(vel.cpu().numpy(), rv, f"vel [{eng.last_kernel()}]", atol=(fd_atol(rp, dt) if fd else 0.0) + slack(v32, rv))
    # the rollout kernels on the same trajectory, bit-exact
    pg, dg = rng.uniform(0.2, 2.0, tc.action_dim), rng.uniform(0.02, 0.3, tc.action_dim)
    n_steps = rng.integers(0, pos.shape[1] + 1, B).astype(np.int32)
    q0, qd0 = rng.uniform(-1, 1, (B, tc.action_dim)), rng.uniform(-0.3, 0.3, (B, tc.action_dim))
    q, qd = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
    spec = RolloutSpec("motor", tc.action_dim, pg, dg, -0.8, 0.8, plant="double_integrator", dt=dt)
    act = eng.pd_rollout(spec, pos, vel, q, qd, n_steps=torch.tensor(n_steps))
    ra, rq, rqd = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -0.8, 0.8, "double_integrator", dt,
                            q0, qd0, n_steps=n_steps)
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
    assert np.array_equal(q.cpu().numpy(), rq) and np.array_equal(qd.cpu().numpy(), rqd)
    if True:
        # one fused launch where the configuration allows it, trajectory + rollout kernels otherwise: same bits
        q2, qd2 = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
        p2, v2, a2 = eng.trajectory_rollout(params, ip, iv, spec, q2, qd2, n_steps=torch.tensor(n_steps),
                                            init_time=init_time)
        assert torch.equal(p2, pos) and torch.equal(v2, vel) and torch.equal(a2, act)
        assert torch.equal(q2, q) and torch.equal(qd2, qd)
    # one replanning step (integer state + plan + rollout + condition gather in one launch where the fused kernel applies)
    # against the separate kernels, from random per-episode integer states
    T = pos.shape[1]
    every, mpt, horizon = int(rng.integers(1, T + 1)), int(rng.integers(1, 5)), int(rng.integers(1, 2 * T + 1))
    ts0 = rng.integers(0, horizon, B).astype(np.int32)
    ps0 = rng.integers(0, 5, B).astype(np.int32)
    dn0 = (rng.random(B) < 0.25).astype(np.uint8)

    def state():
        return (torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda"), torch.tensor(ts0, device="cuda"),
                torch.tensor(ps0, device="cuda"), torch.tensor(dn0, device="cuda"))
    qa, qda, tsa, psa, dna = state()
    r = eng.replan_step(params, ip, iv, spec, qa, qda, tsa, psa, dna, every, mpt, horizon, init_time=init_time,
                        condition=True)
    qb, qdb, tsb, psb, dnb = state()
    seg = eng.replan_advance(tsb, psb, dnb, every, mpt, horizon)
    pb, vb, ab = eng.trajectory_rollout(params, ip, iv, spec, qb, qdb, n_steps=seg, init_time=init_time)
    cp, cv = eng.condition_gather(pb, vb, seg)
    assert torch.equal(r["seg_len"], seg) and torch.equal(tsa, tsb) and torch.equal(psa, psb) and torch.equal(dna, dnb)
    assert torch.equal(r["done"], dnb) and torch.equal(r["pos"], pos) and torch.equal(r["vel"], vel)
    assert torch.equal(r["actions"], ab) and torch.equal(qa, qb) and torch.equal(qda, qdb)
    assert torch.equal(r["cond_pos"], cp) and torch.equal(r["cond_vel"], cv)


# ---- round 3: the two new kernel families (k_traj_wide: more than 16 contraction columns; k_traj_flat: whole-trajectory
# images) under the same random configurations, from generators of their own so that the cases above keep their shapes ----
N_CASES_R3 = int(os.environ.get("MPK_FUZZ_CASES_R3", "80"))
START_R3 = int(os.environ.get("MPK_FUZZ_START_R3", "0"))      # soaks in slices: seeds [START, START + CASES)


@pytest.mark.parametrize("seed", range(START_R3, START_R3 + N_CASES_R3))
def test_random_wide_or_flat_configuration_matches_oracle(seed, mpk_option):
    rng = np.random.default_rng(50_000 + seed)
    pc, bc, tc, dt, dur, B, init_time = random_case(rng)
    r3 = np.random.default_rng(99_000 + seed)
    # shared phase only (both kernels are shared-phase kernels): freeze tau / delay
    pc = dataclasses.replace(pc, learn_tau=False, learn_delay=False)
    wide = bool(r3.random() < 0.5)
    only = os.environ.get("MPK_FUZZ_R3_ONLY", "")              # soak slices: "wide" / "flat" / "" (both)
    if (only == "wide" and not wide) or (only == "flat" and wide):
        pytest.skip("slice")
    if wide:
        bc = dataclasses.replace(bc, num_basis=int(r3.integers(17, 72)), num_basis_outside=0)
    flat = int(r3.choice([0, 1]))
    mpk_option("flat", flat)
    if tc.trajectory_generator_type == "prodmp":
        if (dur + init_time) / pc.tau > 5.9:
            init_time = 0.0
        if dur / pc.tau > 5.9:
            pytest.skip("beyond the ProDMP pre-computation range")
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    torch.cuda.synchronize()
    kern = eng.last_kernel()
    if wide:
        assert kern.startswith("k_traj_wide"), kern
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    p32, v32 = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float32)
    if not (np.isfinite(rp).all() and np.isfinite(p32).all()):
        # dozens of Gaussians on an exponential phase: past tau every one of them underflows and the normalisation is 0 / 0
        # -- in the reference's formula as in the oracle; nothing to compare
        pytest.skip("degenerate basis: every Gaussian underflows somewhere on the horizon (NaN in the reference's formula)")

    def slack(r32, r64):
        e = float(np.abs(r32.astype(np.float64) - r64).max()) if r64.size else 0.0
        return e if e > 2e-6 * float(np.abs(r64).max()) else 0.0
    close(pos.cpu().numpy(), rp, f"pos [{kern}]", atol=slack(p32, rp))
    fd = tc.trajectory_generator_type == "promp"
    close(vel.cpu().numpy(), rv, f"vel [{kern}]", atol=(fd_atol(rp, dt) if fd else 0.0) + slack(v32, rv))
    if not wide:
        # the same launch through the other episode-major / tile-major kernels: identical bits
        mpk_option("flat", 1 - flat)
        p2, v2 = eng.trajectory(params, ip, iv, init_time)
        assert torch.equal(pos, p2) and torch.equal(vel, v2), (kern, eng.last_kernel())


# ---- round 4: k_traj_ring (producer waves + store engine, batch tickets) under random shapes and launch geometries ------------------
N_CASES_R4 = int(os.environ.get("MPK_FUZZ_CASES_R4", "60"))
_RING_SEEN = {"cases": 0, "ring": 0, "burst": 0}
START_R4 = int(os.environ.get("MPK_FUZZ_START_R4", "0"))


@pytest.mark.parametrize("seed", range(START_R4, START_R4 + N_CASES_R4))
def test_random_configuration_through_the_ring_kernel_is_bit_identical(seed, mpk_option):
    """promp / prodmp, shared phase, <= 16 columns and DoF: the forced ring launch (random producers / engine waves / groups per
    batch / waves per group / batch order / contraction loop) against the automatic kernels -- trajectory and fused actions, bit
    for bit -- and the oracle; shapes the ring cannot take (its images do not fit the LDS) must fall through to the other kernels"""
    rng = np.random.default_rng(250_000 + seed)
    pc, bc, tc, dt, dur, B, init_time = random_case(rng)
    if tc.trajectory_generator_type == "dmp":
        tc = dataclasses.replace(tc, trajectory_generator_type="promp")
        bc = dataclasses.replace(bc, basis_generator_type="zero_rbf" if rng.random() < 0.5 else "rbf")
        pc = dataclasses.replace(pc, phase_generator_type="linear")
    pc = dataclasses.replace(pc, learn_tau=False, learn_delay=False)
    if tc.trajectory_generator_type == "prodmp":
        if (dur + init_time) / pc.tau > 5.9:
            init_time = 0.0
        if dur / pc.tau > 5.9:
            pytest.skip("beyond the ProDMP pre-computation range")
    r4 = np.random.default_rng(299_000 + seed)
    B = int(r4.choice([1, 2, 3, 5, 17, 64, 257, 1031, 5003]))
    eng = make_engine(pc, bc, tc, dt, dur)
    D = tc.action_dim
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    mpk_option("ring", 0)
    p0, v0 = [t.clone() for t in eng.trajectory(params, ip, iv, init_time)]
    k0 = eng.last_kernel()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    if not np.isfinite(rp).all():
        pytest.skip("degenerate basis")
    p32, v32 = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float32)

    def slack(r32, r64):
        e = float(np.abs(r32.astype(np.float64) - r64).max()) if r64.size else 0.0
        return e if e > 2e-6 * float(np.abs(r64).max()) else 0.0
    close(p0.cpu().numpy(), rp, f"pos [{k0}]", atol=slack(p32, rp))
    opts = dict(ring_np=int(r4.choice([-1, 1, 3, 8, 10])), ring_ns=int(r4.choice([-1, 1, 2, 4])), ring_m=int(r4.choice([-1, 1, 2, 3, 4, 7])),
                ring_parts=int(r4.choice([-1, 1, 2, 5])), ring_dbg=int(r4.choice([0, 0, 4, 16, 32, 36])))
    for k, v in opts.items():
        mpk_option(k, v)
    ring = int(r4.choice([1, 1, 1, 2]))                # 2: the short-lived-workgroup variant kept for A/B
    mpk_option("ring", ring)
    out = (torch.full_like(p0, float("nan")), torch.full_like(v0, float("nan")))
    eng.trajectory(params, ip, iv, init_time, out=out)
    torch.cuda.synchronize()
    k1 = eng.last_kernel()
    _RING_SEEN["cases"] += 1
    _RING_SEEN["ring"] += k1.startswith("k_traj_ring")
    _RING_SEEN["burst"] += k1.startswith("k_traj_burst")
    assert not k0.startswith(("k_traj_ring", "k_traj_burst")), k0
    assert torch.equal(out[0].view(torch.int32), p0.view(torch.int32)) and torch.equal(out[1].view(torch.int32), v0.view(torch.int32)), (k0, k1, ring, opts)
    if D <= 16 and bc.num_basis + 3 <= 16:
        from fancy_gym_amd import RolloutSpec
        ctrl = str(r4.choice(["motor", "position", "velocity"]))
        spec = RolloutSpec(ctrl, D, r4.uniform(0.1, 2.0, D), r4.uniform(0.01, 0.3, D), -1.5, 1.5, plant="static")
        cp, cv = r4.uniform(-1, 1, (B, D)), r4.uniform(-1, 1, (B, D))
        mpk_option("ring", 0)
        a0 = [t.clone() for t in eng.trajectory_actions(params, ip, iv, spec, cp, cv, init_time=init_time)]
        mpk_option("ring", ring)
        a1 = eng.trajectory_actions(params, ip, iv, spec, cp, cv, init_time=init_time)
        torch.cuda.synchronize()
        for x, y in zip(a0, a1):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (eng.last_kernel(), opts, ctrl)


def test_the_ring_fuzz_reached_the_ring_kernel():
    """most random shapes fit the ring's LDS images: the comparison above is not one of a kernel with itself"""
    if _RING_SEEN["cases"] < 20:
        pytest.skip("the ring fuzz did not run in this process")
    assert _RING_SEEN["ring"] >= 0.4 * _RING_SEEN["cases"], _RING_SEEN
    assert _RING_SEEN["burst"] >= 0.05 * _RING_SEEN["cases"], _RING_SEEN


# ---- round 4, second session: the closed loop on the ring (producers, store engine, consumer waves, action writers) --------------------
N_CASES_RC = int(os.environ.get("MPK_FUZZ_CASES_RC", "40"))
START_RC = int(os.environ.get("MPK_FUZZ_START_RC", "0"))


@pytest.mark.parametrize("seed", range(START_RC, START_RC + N_CASES_RC))
def test_random_closed_loop_step_through_the_ring_is_bit_identical(seed, mpk_option):
    """random promp / prodmp shapes the closed-loop ring takes (5 or 7 DoF, <= 8 contraction columns, T D a multiple of 4), random
    batch sizes, executed steps, controllers, launch geometries and batch orders: trajectory_rollout and one replanning step
    through k_traj_ring<.., closed> against the lane-quarter / pipeline kernels, every output and every state bit for bit"""
    from tests.test_gpu_edge_cases import cfg_for
    r = np.random.default_rng(410_000 + seed)
    mp = str(r.choice(["prodmp", "promp"]))
    D = int(r.choice([5, 7]))
    nb = int(r.integers(1, 5)) if mp == "prodmp" else int(r.integers(2, 8))
    T = 4 * int(r.integers(2, 45))                       # 8 .. 176 steps: T * D is a multiple of 4
    pc, bc, tc, dt, dur = cfg_for(mp, D, nb, T)
    eng = make_engine(pc, bc, tc, dt, dur)
    B = int(r.choice([1, 2, 7, 8, 9, 63, 130, 1025, 2311, 9001]))
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    q0, qd0 = r.uniform(-1, 1, (B, D)), r.uniform(-0.3, 0.3, (B, D))
    ctrl = str(r.choice(["motor", "motor", "position", "velocity"]))
    spec = RolloutSpec(ctrl, D, r.uniform(0.1, 2.0, D), r.uniform(0.01, 0.3, D), -1.2, 1.2, plant="double_integrator", dt=dt)
    n_steps = torch.tensor(r.integers(0, T + 1, B).astype(np.int32)) if r.random() < 0.7 else None
    init_time = float(r.choice([0.0, 0.3 * dur]))
    if mp == "prodmp" and (dur + init_time) / pc.tau > 5.9:
        init_time = 0.0
    opts = dict(ring_np=int(r.choice([-1, 1, 3, 8, 9])), ring_ns=int(r.choice([-1, 1, 2, 3])), ring_nc=int(r.choice([-1, 1, 2, 3, 4, 6])),
                ring_m=int(r.choice([-1, -1, 1, 2, 3])), ring_dbg=int(r.choice([0, 0, 4, 8, 12, 16, 24, 32, 40, 64])),
                write_through=int(r.choice([-1, -1, 0, 1])))

    def run():
        q, qd = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
        out = eng.trajectory_rollout(params, ip, iv, spec, q, qd, n_steps=n_steps, init_time=init_time)
        torch.cuda.synchronize()
        return [x.clone() for x in out] + [q, qd]
    mpk_option("ring", 0)
    ref = run()
    k0 = eng.last_kernel()
    assert "k_traj_ring" not in k0, k0
    mpk_option("ring", 1)
    for k, v in opts.items():
        mpk_option(k, v)
    got = run()
    k1 = eng.last_kernel()
    assert k1.endswith("closed>"), k1
    took = k1.startswith("k_traj_ring")             # (horizons whose images leave no two batch buffers in the LDS fall through)
    assert took or T * D > 700, (k1, mp, D, nb, T)
    for i, (x, y) in enumerate(zip(got, ref)):
        assert torch.equal(x, y), (i, k0, k1, mp, D, nb, T, B, opts)
    # one replanning step with random integer state and the boundary-condition gather
    every, mpt, horizon = int(r.integers(1, T + 1)), int(r.integers(1, 5)), int(r.integers(1, 3 * T))
    ts0 = r.integers(0, horizon, B).astype(np.int32); ps0 = r.integers(0, 4, B).astype(np.int32)
    dn0 = (r.random(B) < 0.2).astype(np.uint8)
    res = []
    for ring in (1, 0):
        mpk_option("ring", ring)
        st = (torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda"), torch.tensor(ts0, device="cuda"),
              torch.tensor(ps0, device="cuda"), torch.tensor(dn0, device="cuda"))
        rr = eng.replan_step(params, ip, iv, spec, *st, every, mpt, horizon, init_time=0.0, condition=bool(seed % 2))
        torch.cuda.synchronize()
        assert ("k_traj_ring" in eng.last_kernel()) == (ring == 1 and took), eng.last_kernel()
        keys = ("pos", "vel", "actions", "seg_len", "done") + (("cond_pos", "cond_vel") if seed % 2 else ())
        res.append([rr[k].clone() for k in keys] + list(st))
    for i, (x, y) in enumerate(zip(*res)):
        assert torch.equal(x, y), (i, "replan", mp, D, nb, T, B, opts)


# ---- round 3, part two: the entry points the two tests above do not reach -- fused open-loop actions (every controller),
# the reacher rollout + reward, validity + penalty, per-episode init_time vectors ------------------------------------------
N_CASES_R3B = int(os.environ.get("MPK_FUZZ_CASES_R3B", "60"))
START_R3B = int(os.environ.get("MPK_FUZZ_START_R3B", "0"))


@pytest.mark.parametrize("seed", range(START_R3B, START_R3B + N_CASES_R3B))
def test_random_actions_reacher_validity_and_per_episode_init_time(seed, mpk_option):
    rng = np.random.default_rng(150_000 + seed)
    pc, bc, tc, dt, dur, B, init_time = random_case(rng)
    if tc.trajectory_generator_type == "prodmp":
        tau_min = pc.tau_bound[0] if pc.learn_tau else pc.tau
        if (dur + init_time) / tau_min > 5.9:
            init_time = 0.0
        if dur / tau_min > 5.9:
            pytest.skip("beyond the ProDMP pre-computation range")
    mpk_option("mapping", int(rng.choice([-1, 1, 2])))
    mpk_option("flat", int(rng.choice([-1, 0, 1])))
    mpk_option("pd_quad", int(rng.choice([-1, 0, 2])))
    mpk_option("pd_simple", int(rng.choice([-1, 0, 1])))
    r3 = np.random.default_rng(99_000 + seed)       # (a generator of its own: the cases keep their shapes)
    mpk_option("phase_flat", int(r3.choice([-1, 0, 1])))
    mpk_option("phase_chunk", int(r3.choice([-1, 1, 2, 3, 5, 7, 8])))
    eng = make_engine(pc, bc, tc, dt, dur)
    D = tc.action_dim
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    if pc.learn_tau:
        params[:, 0] = rng.uniform(pc.tau_bound[0], pc.tau_bound[1], B)
    if pc.learn_delay:
        params[:, int(pc.learn_tau)] = rng.uniform(pc.delay_bound[0], pc.delay_bound[1], B)
    pos, vel = eng.trajectory(params, ip, iv, init_time)
    T = pos.shape[1]
    p_np, v_np = pos.cpu().numpy(), vel.cpu().numpy()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float64)
    if not np.isfinite(rp).all():
        pytest.skip("degenerate basis")
    # (1) trajectory + open-loop actions in one launch (frozen state), every controller: same trajectory bits, actions bit-exact
    ctrl = str(rng.choice(["motor", "position", "velocity"]))
    pg, dg = rng.uniform(0.2, 2.0, D), rng.uniform(0.02, 0.3, D)
    cp, cv = rng.uniform(-1, 1, (B, D)), rng.uniform(-0.3, 0.3, (B, D))
    lo, hi = -float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))
    shared = not (pc.learn_tau or pc.learn_delay)
    if shared:        # (DMP since round 5: one launch on the response route, two launches where that does not apply -- same bits)
        spec = RolloutSpec(ctrl, D, pg, dg, lo, hi, plant="static")
        p2, v2, a2 = eng.trajectory_actions(params, ip, iv, spec, cp, cv, init_time=init_time)
        assert torch.equal(p2, pos) and torch.equal(v2, vel), eng.last_kernel()
        ra, _, _ = O.rollout(p_np, v_np, ctrl, pg, dg, lo, hi, "static", dt, cp, cv)
        assert np.array_equal(a2.cpu().numpy(), ra.astype(np.float32)), eng.last_kernel()
    # (2) reacher rollout + reward on the same desired trajectory
    goal = rng.uniform(-D, D, (B, 2))
    n_steps = rng.integers(0, T + 1, B).astype(np.int32)
    step0 = rng.integers(0, 260, B).astype(np.int32)
    q0, qd0 = rng.uniform(-1, 1, (B, D)), rng.uniform(-0.5, 0.5, (B, D))
    spec_r = RolloutSpec(ctrl, D, pg, dg, lo, hi, plant="double_integrator", dt=dt)
    q, qd = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
    act, rew = eng.reacher_rollout(spec_r, pos, vel, q, qd, torch.tensor(goal), n_steps=torch.tensor(n_steps),
                                   step0=torch.tensor(step0))
    ra, rr, rq, rqd = O.reacher_rollout(p_np, v_np, ctrl, pg, dg, lo, hi, dt, q0, qd0, goal, n_steps=n_steps, step0=step0)
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
    assert np.array_equal(q.cpu().numpy(), rq) and np.array_equal(qd.cpu().numpy(), rqd)
    got = rew.cpu().numpy()
    assert np.all(np.abs(got - rr) <= 1e-11 * (1.0 + np.abs(rr))), np.abs(got - rr).max()
    # (2b) round 5: the same plan + rollout + reward + aggregation in ONE launch that stores nothing per step (mpk_episode_return):
    # plant state and aggregated reward bit for bit as the separate launches (where the fused kernel applies: shared phase, <= 16 columns)
    if shared and D >= 2:
        agg = str(rng.choice(["sum", "mean", "last"]))
        q2, qd2 = torch.tensor(q0, device="cuda"), torch.tensor(qd0, device="cuda")
        try:
            r = eng.episode_return(params, ip, iv, spec_r, q2, qd2, n_steps=torch.tensor(n_steps), reward="simple_reacher",
                                   goal=torch.tensor(goal), step0=torch.tensor(step0), aggregation=agg, init_time=init_time)
        except NotImplementedError:
            r = None
        if r is not None:
            assert eng.last_kernel().startswith("k_episode_return<"), eng.last_kernel()
            want = eng.reward_aggregate(rew, torch.tensor(n_steps), agg)
            # (bit for bit where the step rewards came from the tile kernel -- tests/test_gpu_blackbox.py --; shapes that take the generic
            # rollout kernel (T D not a multiple of 4) add the control cost over the DoF lanes in tree order: last-bit differences)
            assert torch.all((r["ret"] - want).abs() <= 1e-12 * (1.0 + want.abs())), (agg, eng.last_kernel(), (r["ret"] - want).abs().max())
            assert torch.equal(q2, q) and torch.equal(qd2, qd), eng.last_kernel()
    # (3) validity + the penalty of an invalid plan
    lim = np.sort(rng.uniform(-2.0, 2.0, (2, D)), axis=0)
    p64 = p_np.astype(np.float64)
    want = np.all((p64 >= lim[0]) & (p64 <= lim[1]), axis=(1, 2))
    got_v, pen = eng.traj_validity(pos, lim[0], lim[1], with_penalty=True)
    assert np.array_equal(got_v.cpu().numpy(), want)
    assert np.allclose(pen.cpu().numpy(), O.traj_invalid_penalty(params, p64, lim[0], lim[1]), rtol=1e-12, atol=1e-15)
    # (4) per-episode init_time vector (all equal): the per-episode-phase kernels against the same oracle trajectory
    if T >= 2:
        itv = torch.full((B,), float(init_time), dtype=torch.float32, device="cuda")
        p3, v3 = eng.trajectory(params, ip, iv, itv)
        p32, v32 = O.get_trajectory(pc, bc, tc, params, dur, dt, init_time, ip, iv, dtype=np.float32)

        def slack(r32, r64):
            e = float(np.abs(r32.astype(np.float64) - r64).max()) if r64.size else 0.0
            return e if e > 2e-6 * float(np.abs(r64).max()) else 0.0
        kern = eng.last_kernel()
        close(p3.cpu().numpy(), rp, f"pos [{kern}]", atol=slack(p32, rp))
        fd = tc.trajectory_generator_type == "promp"
        close(v3.cpu().numpy(), rv, f"vel [{kern}]", atol=(fd_atol(rp, dt) if fd else 0.0) + slack(v32, rv))


# ---- part three: whole BatchedBlackBox episodes (plan -> execute -> re-condition -> replan) --------------------------------
N_CASES_BB = int(os.environ.get("MPK_FUZZ_CASES_BB", "60"))
START_BB = int(os.environ.get("MPK_FUZZ_START_BB", "0"))


def _batched_from_cfg(pc, bc, tc, dt, dur, B, ctrl, pg, dg, lo, hi, **kw):
    from fancy_gym_amd import BatchedBlackBox
    from fancy_gym_amd.black_box.factory import (get_basis_generator, get_controller, get_phase_generator,
                                                 get_trajectory_generator)
    pkw = dict(tau=pc.tau, delay=pc.delay, learn_tau=pc.learn_tau, learn_delay=pc.learn_delay,
               tau_bound=list(pc.tau_bound), delay_bound=list(pc.delay_bound))
    if pc.phase_generator_type == "exp":
        pkw["alpha_phase"] = pc.alpha_phase
    phase = get_phase_generator(pc.phase_generator_type, **pkw)
    bkw = dict(num_basis=bc.num_basis, basis_bandwidth_factor=bc.basis_bandwidth_factor)
    if bc.basis_generator_type == "zero_rbf":
        bkw.update(num_basis_zero_start=bc.num_basis_zero_start, num_basis_zero_goal=bc.num_basis_zero_goal)
    else:
        bkw.update(num_basis_outside=bc.num_basis_outside)
    if bc.basis_generator_type == "prodmp":
        bkw.update(alpha=bc.alpha, dt=bc.dt)
    basis = get_basis_generator(bc.basis_generator_type, phase, **bkw)
    tkw = dict(weights_scale=tc.weights_scale)
    if tc.trajectory_generator_type == "prodmp":
        tkw.update(goal_scale=tc.goal_scale, auto_scale_basis=tc.auto_scale_basis, relative_goal=tc.relative_goal,
                   disable_goal=tc.disable_goal, disable_weights=tc.disable_weights)
    elif tc.trajectory_generator_type == "dmp":
        tkw.update(goal_scale=tc.goal_scale, alpha=tc.alpha)
    tg = get_trajectory_generator(tc.trajectory_generator_type, tc.action_dim, basis, **tkw)
    ckw = {} if ctrl != "motor" else dict(p_gains=pg, d_gains=dg)
    return BatchedBlackBox(tg, get_controller(ctrl, **ckw), B, dt, dur, act_low=lo, act_high=hi, **kw)


@pytest.mark.parametrize("seed", range(START_BB, START_BB + N_CASES_BB))
def test_random_batched_episode_follows_the_oracle_sequence(seed, mpk_option):
    """
    BatchedBlackBox.step over whole episodes under a random configuration, replanning schedule `t % every == 0`,
    max_planning_times, condition_on_desired on / off, controller, action limits and kernel options -- the fused step
    (mpk_replan_step) and the unfused one (trajectory, integer rule, rollout, condition gather as separate launches) on two
    instances must agree bit for bit, and both follow the oracle's sequence: plans within 1e-5, the integer state, actions and
    plant state bit-exact given the plan (black_box_wrapper.py:150-217, test/test_replanning_sequencing.py).
    """
    rng = np.random.default_rng(123_000 + seed)
    pc, bc, tc, dt, dur, B, _ = random_case(rng)
    B = min(B, 33)
    D = tc.action_dim
    T = int(round(dur / dt))
    if T < 2:
        pytest.skip("one-step horizon")
    replan = bool(rng.random() < 0.75)
    every = int(rng.integers(1, T + 1)) if replan else None
    mpt = [1, 2, 3, 5, float("inf")][int(rng.integers(0, 5))]
    cod = bool(rng.random() < 0.5)
    if tc.trajectory_generator_type == "prodmp" and 2.0 * dur / min(pc.tau, pc.tau_bound[0] if pc.learn_tau else pc.tau) > 5.9:
        pytest.skip("beyond the ProDMP pre-computation range")
    ctrl = str(rng.choice(["motor", "position", "velocity"]))
    pg, dg = rng.uniform(0.2, 2.0, D), rng.uniform(0.02, 0.3, D)
    lo, hi = -float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))
    for key, vals in (("mapping", [-1, 1, 2]), ("quad", [-1, 0, 2, 3, 4]), ("bulk", [-1, 0, 2]), ("pipe", [-1, 0, 1]),
                      ("split", [-1, 0, 1]), ("pd_quad", [-1, 0, 2]), ("pd_simple", [-1, 0, 1]), ("phase_flat", [-1, 0, 1]),
                      ("phase_chunk", [-1, 1, 2, 3, 4, 7])):
        mpk_option(key, int(rng.choice(vals)))
    r6 = np.random.default_rng(66_000 + seed)          # (round 6; own generator: cases keep their shapes)
    mpk_option("phase_pipe", int(r6.choice([-1, 0, 1])))
    mpk_option("pd_pipe", int(r6.choice([-1, 0, 1])))
    kw = dict(plant="double_integrator", max_planning_times=mpt, condition_on_desired=cod)
    if replan:
        kw["replanning_every"] = every
    fused = _batched_from_cfg(pc, bc, tc, dt, dur, B, ctrl, pg, dg, lo, hi, **kw)
    plain = _batched_from_cfg(pc, bc, tc, dt, dur, B, ctrl, pg, dg, lo, hi, **kw)
    P = fused.engine.num_params
    n_phase = int(pc.learn_tau) + int(pc.learn_delay)
    q0, qd0 = rng.uniform(-1, 1, (B, D)), rng.uniform(-0.5, 0.5, (B, D))
    fused.reset(q0, qd0); plain.reset(q0, qd0)
    q, qd = q0.copy(), qd0.copy()
    cond_p, cond_v = q0.astype(np.float32), qd0.astype(np.float32)
    segments = O.replanning_segments(T, every, mpt) if replan else [(0, T)]
    frozen = None
    for k, (start, n) in enumerate(segments):
        params = (rng.standard_normal((B, P)) * 0.7).astype(np.float32)
        if pc.learn_tau:
            params[:, 0] = rng.uniform(0.8 * pc.tau_bound[0], 1.1 * pc.tau_bound[1], B)
        if pc.learn_delay:
            params[:, int(pc.learn_tau)] = rng.uniform(-0.05 * dur, 1.1 * pc.delay_bound[1] + 1e-3, B)
        of, op = fused.step(params, fuse=True), plain.step(params, fuse=False)
        torch.cuda.synchronize()
        for key in ("des_pos", "des_vel", "step_actions", "trajectory_length", "done"):
            assert torch.equal(of[key], op[key]), (k, key, fused.engine.last_kernel(), plain.engine.last_kernel())
        assert torch.equal(fused.q, plain.q) and torch.equal(fused.qd, plain.qd), k
        assert torch.equal(fused.traj_steps, plain.traj_steps) and torch.equal(fused.plan_steps, plain.plan_steps)
        # the oracle's sequence
        assert torch.all(of["trajectory_length"] == n) and int(fused.traj_steps[0]) == start + n, (k, start, n)
        assert int(fused.plan_steps[0]) == k + 1 and bool(of["done"].all()) == (k == len(segments) - 1)
        p_eff = params.copy()
        if n_phase:
            if frozen is None:       # tau / delay are frozen (clipped) by the first plan of an episode
                bounds = ([pc.tau_bound] if pc.learn_tau else []) + ([pc.delay_bound] if pc.learn_delay else [])
                frozen = np.stack([np.clip(params[:, i], np.float32(b[0]), np.float32(b[1])) for i, b in enumerate(bounds)], 1)
            p_eff[:, :n_phase] = frozen
        rp, rv = O.get_trajectory(pc, bc, tc, p_eff, dur, dt, start * dt, cond_p, cond_v, dtype=np.float64)
        if not np.isfinite(rp).all():
            pytest.skip("degenerate basis")
        dp, dv = of["des_pos"].cpu().numpy(), of["des_vel"].cpu().numpy()
        p32, v32 = O.get_trajectory(pc, bc, tc, p_eff, dur, dt, start * dt, cond_p, cond_v, dtype=np.float32)

        def slack(r32, r64):
            e = float(np.abs(r32.astype(np.float64) - r64).max()) if r64.size else 0.0
            return e if e > 2e-6 * float(np.abs(r64).max()) else 0.0
        fd = tc.trajectory_generator_type == "promp"
        close(dp, rp, f"plan {k} pos [{fused.engine.last_kernel()}]", atol=slack(p32, rp))
        close(dv, rv, f"plan {k} vel", atol=(fd_atol(rp, dt) if fd else 0.0) + slack(v32, rv))
        ra, q, qd = O.rollout(dp, dv, ctrl, pg, dg, lo, hi, "double_integrator", dt, q, qd, n_steps=np.full(B, n))
        assert np.array_equal(of["step_actions"].cpu().numpy(), ra.astype(np.float32)), k
        assert np.array_equal(fused.q.cpu().numpy(), q) and np.array_equal(fused.qd.cpu().numpy(), qd), k
        if cod:
            cond_p, cond_v = dp[:, n - 1], dv[:, n - 1]
        else:
            cond_p, cond_v = q.astype(np.float32), qd.astype(np.float32)


# ---- part four: the validity gate -- episodes that end early and the others drifting out of lockstep -------------------------
N_CASES_GATE = int(os.environ.get("MPK_FUZZ_CASES_GATE", "40"))
START_GATE = int(os.environ.get("MPK_FUZZ_START_GATE", "0"))


@pytest.mark.parametrize("seed", range(START_GATE, START_GATE + N_CASES_GATE))
def test_random_batched_episode_with_validity_gate(seed, mpk_option):
    """
    BatchedBlackBox with position limits (table_tennis_env.py:303-309 on the batched path): a plan that leaves the limits
    ends ITS episode without a plant step (black_box_wrapper.py:169-172), the others go on -- with per-episode init_time
    once they no longer move in lockstep.  Against a per-episode replay of the same rules on the oracle: validity decided on
    the plan the device produced, integer state, actions and plant state bit-exact given that plan, plans within 1e-5.
    """
    rng = np.random.default_rng(321_000 + seed)
    pc, bc, tc, dt, dur, B, _ = random_case(rng)
    B = min(B, 16)
    D = tc.action_dim
    T = int(round(dur / dt))
    if T < 2:
        pytest.skip("one-step horizon")
    if tc.trajectory_generator_type == "prodmp" and 2.0 * dur / min(pc.tau, pc.tau_bound[0] if pc.learn_tau else pc.tau) > 5.9:
        pytest.skip("beyond the ProDMP pre-computation range")
    every = int(rng.integers(max(1, T // 6), T + 1))           # at most ~7 plans per episode
    mpt = [2, 3, 5, float("inf")][int(rng.integers(0, 4))]
    cod = bool(rng.random() < 0.5)
    ctrl = str(rng.choice(["motor", "position", "velocity"]))
    pg, dg = rng.uniform(0.2, 2.0, D), rng.uniform(0.02, 0.3, D)
    lo, hi = -float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))
    for key, vals in (("phase_flat", [-1, 0, 1]), ("phase_chunk", [-1, 1, 2, 3, 4, 7]), ("phase", [-1, 0, 1]),
                      ("phase_table", [-1, 0, 1]), ("pd_quad", [-1, 0, 2]), ("pd_simple", [-1, 0, 1]), ("mapping", [-1, 1, 2])):
        mpk_option(key, int(rng.choice(vals)))
    r6 = np.random.default_rng(66_000 + seed)       # round 6: the gate in the producer / consumer kernels (own generator: cases keep their shapes)
    mpk_option("phase_pipe", int(r6.choice([-1, 0, 1])))
    mpk_option("pipe", int(r6.choice([-1, 0, 1])))
    L = float(rng.uniform(0.8, 2.5))
    limits = (np.full(D, -L), np.full(D, L * float(rng.uniform(0.7, 1.3))))
    bb = _batched_from_cfg(pc, bc, tc, dt, dur, B, ctrl, pg, dg, lo, hi, plant="double_integrator", replanning_every=every,
                           max_planning_times=mpt, condition_on_desired=cod, pos_limits=limits)
    P = bb.engine.num_params
    n_phase = int(pc.learn_tau) + int(pc.learn_delay)
    q0, qd0 = rng.uniform(-0.6, 0.6, (B, D)), rng.uniform(-0.3, 0.3, (B, D))
    bb.reset(q0, qd0)
    # per-episode oracle state
    q, qd = q0.copy(), qd0.copy()
    cond_p, cond_v = q0.astype(np.float32), qd0.astype(np.float32)
    steps, plans, done = np.zeros(B, np.int64), np.zeros(B, np.int64), np.zeros(B, bool)
    frozen = None
    for k in range(12):
        if done.all():
            break
        params = (rng.standard_normal((B, P)) * 0.5).astype(np.float32)
        if pc.learn_tau:
            params[:, 0] = rng.uniform(pc.tau_bound[0], pc.tau_bound[1], B)
        if pc.learn_delay:
            params[:, int(pc.learn_tau)] = rng.uniform(pc.delay_bound[0], pc.delay_bound[1], B)
        was_done = done.copy()
        out = bb.step(params)
        torch.cuda.synchronize()
        dp, dv = out["des_pos"].cpu().numpy(), out["des_vel"].cpu().numpy()
        act = out["step_actions"].cpu().numpy()
        seg_got = out["trajectory_length"].cpu().numpy()
        p_eff = params.copy()
        if n_phase:
            if frozen is None:
                bounds = ([pc.tau_bound] if pc.learn_tau else []) + ([pc.delay_bound] if pc.learn_delay else [])
                frozen = np.stack([np.clip(params[:, i], np.float32(b[0]), np.float32(b[1])) for i, b in enumerate(bounds)], 1)
            p_eff[:, :n_phase] = frozen
        valid = np.all((dp.astype(np.float64) >= limits[0]) & (dp.astype(np.float64) <= limits[1]), axis=(1, 2))
        assert np.array_equal(out["valid"].cpu().numpy(), valid), k
        for b in range(B):
            if was_done[b]:
                assert seg_got[b] == 0 and np.all(act[b] == 0), (k, b)
                continue
            it = float(np.float32(steps[b] * dt))
            rp, rv = O.get_trajectory(pc, bc, tc, p_eff[b:b + 1], dur, dt, it, cond_p[b:b + 1], cond_v[b:b + 1], dtype=np.float64)
            if not np.isfinite(rp).all():
                pytest.skip("degenerate basis")
            p32, v32 = O.get_trajectory(pc, bc, tc, p_eff[b:b + 1], dur, dt, it, cond_p[b:b + 1], cond_v[b:b + 1], dtype=np.float32)
            e = float(np.abs(p32.astype(np.float64) - rp).max())
            # a plan whose terms cancel (seed 83806: two weights of ~0.3 summing to 5e-4 over a saturated phase) carries the
            # fp32 rounding of its TERMS, not of its result: both fp32 evaluations are then ~one ulp of a term off the float64
            # one, each in its own direction
            terms = float(np.abs(p_eff[b]).max()) * max(tc.weights_scale, tc.goal_scale, 1.0) + float(np.abs(cond_p[b]).max())
            slack = (2.0 * e if e > 2e-6 * float(np.abs(rp).max()) else 0.0) + 2.0 * float(np.finfo(np.float32).eps) * terms
            close(dp[b:b + 1], rp, f"plan {k} episode {b} pos [{bb.engine.last_kernel()}]", atol=slack)
            if not valid[b]:
                done[b] = True
                assert seg_got[b] == 0 and np.all(act[b] == 0), (k, b)
                continue
            plans[b] += 1
            g_break = T
            if plans[b] < mpt:
                g_break = min((steps[b] // every + 1) * every, T)
            n = int(max(1, min(g_break - steps[b], T)))
            assert seg_got[b] == n, (k, b, seg_got[b], n)
            ra, qb, qdb = O.rollout(dp[b:b + 1], dv[b:b + 1], ctrl, pg, dg, lo, hi, "double_integrator", dt, q[b:b + 1], qd[b:b + 1],
                                    n_steps=np.full(1, n))
            q[b], qd[b] = qb[0], qdb[0]
            assert np.array_equal(act[b], ra[0].astype(np.float32)), (k, b)
            steps[b] += n
            done[b] = steps[b] >= T
            if cod:
                cond_p[b], cond_v[b] = dp[b, n - 1], dv[b, n - 1]
            else:
                cond_p[b], cond_v[b] = q[b].astype(np.float32), qd[b].astype(np.float32)
        assert np.array_equal(bb.q.cpu().numpy(), q) and np.array_equal(bb.qd.cpu().numpy(), qd), k
        assert np.array_equal(bb.traj_steps.cpu().numpy(), steps) and np.array_equal(bb.plan_steps.cpu().numpy(), plans), k
        assert np.array_equal(out["done"].cpu().numpy(), done), k
    assert done.all()
