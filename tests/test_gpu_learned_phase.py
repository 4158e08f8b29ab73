"""
The reference's LEARNED-PHASE configurations on the one-launch path (round 6, k_phase_fused: csrc/mpk_phase_fused.hip):
    TableTennis-ProDMP          envs/mujoco/table_tennis/mp_wrapper.py:32-57   learn tau + delay, 7 x 3, alpha 25, T = 350
    TableTennis-ProDMP Replan   :91-121                                       7 x 2 (+ goal), t % 50 == 0, max_planning_times 3
    BeerPong-ProMP              envs/mujoco/beerpong/mp_wrapper.py:9-25        learn tau, 7 x 2 + 2 zero-start, T = 300
mpk_trajectory_actions / mpk_trajectory_rollout / mpk_replan_step(_gated) / mpk_episode_return(_gated) must be ONE launch for them and
equal -- bit for bit -- what the separate launches give (mpk_trajectory + mpk_pd_rollout + mpk_replan_advance + mpk_condition_gather
+ mpk_traj_validity_penalty), which in turn are checked against the oracle: trajectories 1e-5 (fp32 AND fp64 oracle), actions / plant
state / integer state exactly from the reference's loop (oracle.rollout, oracle.replanning_segments).
"""
import os

import numpy as np
import pytest
import torch

from fancy_gym_amd import RolloutSpec, _lib
from oracle import mp_oracle as O
from tests.test_gpu_trajectory import close, fd_atol, make_engine

pytestmark = pytest.mark.gpu

TT_P = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
TT_D = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])
BP_P = np.array([1.5, 5, 2.55, 3, 2., 2, 1.25])
BP_D = np.array([0.02333333, 0.1, 0.0625, 0.08, 0.03, 0.03, 0.0125])
JNT_LOW = np.array([-2.6, -2.0, -2.8, -0.9, -4.8, -1.6, -2.2])       # table_tennis_utils.py:3-4
JNT_HIGH = np.array([2.6, 2.0, 2.8, 3.1, 1.3, 1.6, 2.2])
TT_PHASE = dict(tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5), delay_bound=(0.05, 0.15))

# (phase, basis, trajectory generator, dt, duration, gains, replanning every, max_planning_times)
CONFIGS = {
    "tt_prodmp": (O.PhaseCfg("exp", **TT_PHASE), O.BasisCfg("prodmp", num_basis=3, alpha=25, basis_bandwidth_factor=3),
                  O.TrajCfg("prodmp", action_dim=7, weights_scale=0.7, auto_scale_basis=True, relative_goal=True, disable_goal=True),
                  0.008, 2.8, (TT_P, TT_D), None, 1),
    "tt_prodmp_replan": (O.PhaseCfg("exp", **TT_PHASE), O.BasisCfg("prodmp", num_basis=2, alpha=25, basis_bandwidth_factor=3),
                         O.TrajCfg("prodmp", action_dim=7, auto_scale_basis=True, goal_offset=1.0), 0.008, 2.8, (TT_P, TT_D), 50, 3),
    "beerpong_promp": (O.PhaseCfg("linear", tau=3.0, learn_tau=True, tau_bound=(0.02, 3.0)),
                       O.BasisCfg("zero_rbf", num_basis=2, num_basis_zero_start=2, num_basis_zero_goal=0, basis_bandwidth_factor=3),
                       O.TrajCfg("promp", action_dim=7), 0.01, 3.0, (BP_P, BP_D), None, 1),
    # shapes beside the reference's: other DoF counts (run-time DoF loop), eight ProMP columns, a horizon that is no multiple of 16 or 4
    "promp_5dof_learn_both": (O.PhaseCfg("linear", tau=2.0, learn_tau=True, learn_delay=True, tau_bound=(0.5, 2.0), delay_bound=(0.0, 0.3)),
                              O.BasisCfg("zero_rbf", num_basis=6, num_basis_zero_start=1, num_basis_zero_goal=1, basis_bandwidth_factor=3),
                              O.TrajCfg("promp", action_dim=5), 0.02, 1.5, (1.0, 0.1), 20, 4),
    "prodmp_3dof_learn_tau": (O.PhaseCfg("exp", tau=1.0, alpha_phase=3.0, learn_tau=True, tau_bound=(0.5, 1.2)),
                              O.BasisCfg("prodmp", num_basis=4, alpha=10, basis_bandwidth_factor=2),
                              O.TrajCfg("prodmp", action_dim=3, weights_scale=0.5, goal_scale=2.0), 0.02, 0.86, (2.0, 0.3), 10, 2),
    "promp_16dof_learn_tau": (O.PhaseCfg("linear", tau=1.0, learn_tau=True, tau_bound=(0.3, 1.0)),
                              O.BasisCfg("rbf", num_basis=3, basis_bandwidth_factor=3), O.TrajCfg("promp", action_dim=16), 0.02, 1.0,
                              (1.0, 0.1), None, 1),
}
REFERENCE = ["tt_prodmp", "tt_prodmp_replan", "beerpong_promp"]
# shared-phase configurations for the validity gate's other home, the lane-quarter kernels (gate_pass, csrc/mpk_tile.h): BASELINE cfg5 -- the
# flow the gate exists for (table_tennis/mp_wrapper.py:11-30, table_tennis_env.py:282-309) --, cfg4 (replanning, condition_on_desired) and a
# DMP on its response route
SHARED = {
    "cfg5_promp_tabletennis": (O.PhaseCfg("linear", tau=2.8),
                               O.BasisCfg("zero_rbf", num_basis=3, num_basis_zero_start=1, num_basis_zero_goal=1, basis_bandwidth_factor=3),
                               O.TrajCfg("promp", action_dim=7), 0.008, 2.8, (TT_P, TT_D), None, 1),
    "cfg4_prodmp_replan": (O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0), O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=3, alpha=10),
                           O.TrajCfg("prodmp", action_dim=7, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True, disable_goal=True),
                           0.02, 2.0, (0.01 * np.array([120., 120., 120., 120., 50., 30., 10.]), 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])), 25, 4),
    "dmp_5dof_response": (O.PhaseCfg("exp", tau=2.0, alpha_phase=2.0), O.BasisCfg("rbf", num_basis=5, basis_bandwidth_factor=3),
                          O.TrajCfg("dmp", action_dim=5, alpha=25.0), 0.02, 2.0, (1.0, 0.1), 40, 3),
}
CONFIGS_ALL = dict(CONFIGS, **SHARED)


def make_inputs(name, B, seed=0, scale=1.0):
    pc, bc, tc = CONFIGS_ALL[name][:3]
    rng = np.random.default_rng(seed)
    P = O.num_params(pc, bc, tc)
    params = (scale * rng.standard_normal((B, P))).astype(np.float32)
    i = 0
    if pc.learn_tau:
        lo, hi = pc.tau_bound
        params[:, i] = rng.uniform(max(lo, 0.2 * hi) - 0.05, hi + 0.05, B); i += 1     # (some outside the bounds: the kernels clip)
    if pc.learn_delay:
        lo, hi = pc.delay_bound
        params[:, i] = rng.uniform(lo - 0.02, hi + 0.02, B); i += 1
    ip = rng.uniform(-1, 1, (B, tc.action_dim)).astype(np.float32)
    iv = rng.uniform(-1, 1, (B, tc.action_dim)).astype(np.float32)
    return params, ip, iv


def engine_of(name, **kw):
    pc, bc, tc, dt, dur = CONFIGS_ALL[name][:5]
    return make_engine(pc, bc, tc, dt, dur, device=0, **kw)


def specs(name, low=-1.0, high=1.0, ctrl="motor"):
    pc, bc, tc, dt, dur, (pg, dg) = CONFIGS_ALL[name][:6]
    D = tc.action_dim
    return (RolloutSpec(ctrl, D, pg, dg, low, high, plant="static"),
            RolloutSpec(ctrl, D, pg, dg, low, high, plant="double_integrator", dt=dt))


def cu(x, dtype=None):
    t = torch.as_tensor(x, device="cuda")
    return t if dtype is None else t.to(dtype)


def eq(a, b, what):
    a, b = np.asarray(a.cpu() if torch.is_tensor(a) else a), np.asarray(b.cpu() if torch.is_tensor(b) else b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), f"{what}: {int((a != b).sum())} of {a.size} entries differ (max |diff| {np.abs(a.astype(np.float64) - b.astype(np.float64)).max():.3e})"


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("B", [1, 3, 64, 1000])
def test_fused_actions_are_one_launch_and_equal_the_two_launches_and_the_oracle(name, B):
    pc, bc, tc, dt, dur, (pg, dg) = CONFIGS[name][:6]
    eng = engine_of(name)
    params, ip, iv = make_inputs(name, B)
    static, _ = specs(name)
    cp, cv = ip.astype(np.float64) * 0.5, iv.astype(np.float64) * 0.25
    pos, vel, act = eng.trajectory_actions(params, ip, iv, static, cp, cv)
    assert eng.last_kernel().startswith("k_phase_fused<") and eng.last_kernel().endswith("act>"), eng.last_kernel()
    # the trajectory-only launch + the rollout on it: the same bits
    pos2, vel2 = eng.trajectory(params, ip, iv, 0.0)
    act2 = eng.pd_rollout(static, pos2, vel2, cu(cp), cu(cv))
    torch.cuda.synchronize()
    eq(pos, pos2, "pos"); eq(vel, vel2, "vel"); eq(act, act2, "actions")
    # the oracle: trajectories to 1e-5 (both precisions), actions from ITS loop on the GPU's trajectories exactly.  Against the fp32
    # oracle the tolerance grows by that oracle's own distance from the float64 one where that exceeds 2e-6 of the scale ("within 1e-5
    # of the reference" cannot be resolved finer than the reference's own fp32 error: DESIGN section 3, tests/test_gpu_fuzz.py)
    rp64, rv64 = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=np.float64)
    for dtype in (np.float64, np.float32):
        rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, 0.0, ip, iv, dtype=dtype)
        fd = fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0
        ep = float(np.abs(rp - rp64).max()); ev = float(np.abs(rv - rv64).max())
        close(pos.cpu().numpy(), rp, f"pos {dtype.__name__}", atol=ep if ep > 2e-6 * np.abs(rp64).max() else 0.0)
        close(vel.cpu().numpy(), rv, f"vel {dtype.__name__}", atol=fd + (ev if ev > 2e-6 * np.abs(rv64).max() else 0.0))
    ra, _, _ = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -1.0, 1.0, "static", dt, cp, cv)
    eq(act, ra.astype(np.float32), "actions vs oracle")


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("B", [1, 5, 200, 1000])
def test_fused_closed_loop_equals_the_two_launches_and_the_reference_loop(name, B):
    pc, bc, tc, dt, dur, (pg, dg) = CONFIGS[name][:6]
    eng = engine_of(name)
    T, D = eng.num_steps, eng.num_dof
    params, ip, iv = make_inputs(name, B, seed=1)
    _, closed = specs(name)
    rng = np.random.default_rng(5)
    n_steps = rng.integers(0, T + 1, B).astype(np.int32)
    n_steps[: min(B, 3)] = [T, 0, T - 1][: min(B, 3)]
    for ns in (None, n_steps):
        q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
        q, qd = cu(q0).clone(), cu(qd0).clone()
        pos, vel, act = eng.trajectory_rollout(params, ip, iv, closed, q, qd, n_steps=None if ns is None else cu(ns))
        assert eng.last_kernel().startswith("k_phase_fused<") and "closed" in eng.last_kernel(), eng.last_kernel()
        pos2, vel2 = eng.trajectory(params, ip, iv, 0.0)
        q2, qd2 = cu(q0).clone(), cu(qd0).clone()
        act2 = eng.pd_rollout(closed, pos2, vel2, q2, qd2, n_steps=None if ns is None else cu(ns))
        torch.cuda.synchronize()
        eq(pos, pos2, "pos"); eq(vel, vel2, "vel"); eq(act, act2, "actions"); eq(q, q2, "q"); eq(qd, qd2, "qd")
        ra, rq, rqd = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -1.0, 1.0, "double_integrator", dt, q0, qd0,
                                n_steps=ns)
        eq(act, ra.astype(np.float32), "actions vs oracle"); eq(q, rq, "q vs oracle"); eq(qd, rqd, "qd vs oracle")


def _separate_step(eng, params, cond_pos, cond_vel, closed, q, qd, ts, ps, dn, every, mpt, horizon, init_time, gate=None, raw=None):
    """the launches a step consisted of until round 5 (BatchedBlackBox._step_full)"""
    pos, vel = eng.trajectory(params, cond_pos, cond_vel, init_time)
    out = {}
    if gate is not None:
        chk = gate.get("check_tau_delay", False)
        valid, pen = eng.traj_validity(pos, gate["pos_low"], gate["pos_high"], cu(raw) if chk else None, gate.get("tau_bound") if chk else None,
                                       gate.get("delay_bound") if chk else None, with_penalty=True)
        dn |= (~valid).to(torch.uint8)
        out.update(valid=valid, penalty=pen)
    seg = eng.replan_advance(ts, ps, dn, every, mpt, horizon)
    act = eng.pd_rollout(closed, pos, vel, q, qd, n_steps=seg)
    cp, cv = eng.condition_gather(pos, vel, seg)
    out.update(pos=pos, vel=vel, actions=act, seg_len=seg, cond_pos=cp, cond_vel=cv)
    return out


def _state(B, q0, qd0):
    i32 = dict(dtype=torch.int32, device="cuda")
    return dict(q=cu(q0).clone(), qd=cu(qd0).clone(), ts=torch.zeros(B, **i32), ps=torch.zeros(B, **i32),
                dn=torch.zeros(B, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("lean", [False, True])
@pytest.mark.parametrize("B", [4, 777])
def test_whole_episodes_of_replanning_steps_fused_against_separate_launches_and_the_reference_counters(name, lean, B):
    """every plan of an episode: integer state, plant state, boundary condition of the next plan, actions -- mpk_replan_step (and the
    verbose < 2 step mpk_episode_return) against the separate launches, tau / delay frozen by the first plan"""
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS[name]
    eng = engine_of(name)
    T, D = eng.num_steps, eng.num_dof
    horizon = T
    every_ = every or horizon + 1
    segments = O.replanning_segments(horizon, every_, mpt)
    _, closed = specs(name)
    params0, ip, iv = make_inputs(name, B, seed=2)
    n_ph = int(pc.learn_tau) + int(pc.learn_delay)
    lo_hi = O.params_bounds(pc, bc, tc)
    frozen = np.clip(params0[:, :n_ph], lo_hi[0, :n_ph], lo_hi[1, :n_ph])
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    A, S = _state(B, q0, qd0), _state(B, q0, qd0)
    condA = condS = (cu(ip), cu(iv))
    rng = np.random.default_rng(9)
    for k, (cur, length) in enumerate(segments):
        params = (0.5 * rng.standard_normal(params0.shape)).astype(np.float32)
        params[:, :n_ph] = frozen
        it = cur * dt
        if lean:
            r = eng.episode_return(params, condA[0], condA[1], closed, A["q"], A["qd"], replan=(A["ts"], A["ps"], A["dn"], every_, mpt, horizon),
                                   init_time=it, condition=True)
            assert "lean" in eng.last_kernel(), eng.last_kernel()
            assert float(r["ret"].abs().max()) == 0.0
        else:
            r = eng.replan_step(params, condA[0], condA[1], closed, A["q"], A["qd"], A["ts"], A["ps"], A["dn"], every_, mpt, horizon,
                                init_time=it, condition=True)
            assert eng.last_kernel().startswith("k_phase_fused<") and "closed" in eng.last_kernel(), eng.last_kernel()
        s = _separate_step(eng, params, condS[0], condS[1], closed, S["q"], S["qd"], S["ts"], S["ps"], S["dn"], every_, mpt, horizon, it)
        torch.cuda.synchronize()
        assert int(r["seg_len"].min()) == int(r["seg_len"].max()) == length, (k, length)
        for key in ("q", "qd", "ts", "ps", "dn"):
            eq(A[key], S[key], f"plan {k}: {key}")
        eq(r["seg_len"], s["seg_len"], f"plan {k}: seg_len"); eq(r["done"], S["dn"], f"plan {k}: done snapshot")
        eq(r["cond_pos"], s["cond_pos"], f"plan {k}: cond_pos"); eq(r["cond_vel"], s["cond_vel"], f"plan {k}: cond_vel")
        if not lean:
            eq(r["pos"], s["pos"], f"plan {k}: pos"); eq(r["vel"], s["vel"], f"plan {k}: vel"); eq(r["actions"], s["actions"], f"plan {k}: actions")
            # the reference's loop on this plan, from the state the plan started in
            if k == 0:
                rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, it, ip, iv, dtype=np.float64)
                close(r["pos"].cpu().numpy(), rp, "plan 0 pos");
                close(r["vel"].cpu().numpy(), rv, "plan 0 vel", atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
        condA = (r["cond_pos"], r["cond_vel"]); condS = (s["cond_pos"], s["cond_vel"])
    assert int(A["ts"].min()) == horizon and int(A["dn"].min()) == 1 and int(A["ps"].max()) == len(segments)


@pytest.mark.parametrize("name", ["tt_prodmp", "tt_prodmp_replan", "promp_5dof_learn_both", "cfg5_promp_tabletennis", "cfg4_prodmp_replan",
                                  "dmp_5dof_response"])
@pytest.mark.parametrize("lean", [False, True])
@pytest.mark.parametrize("quad", [None, 2, 3, 4])
def test_the_validity_gate_inside_the_step_equals_the_separate_launches_and_the_oracle(name, lean, quad, mpk_option):
    """joint limits + tau / delay bounds on the RAW action, penalty, and the roll-back of an invalid plan (black_box_wrapper.py:155-172;
    table_tennis_env.py:282-309): mpk_replan_step_gated / mpk_episode_return_gated against trajectory + mpk_traj_validity_penalty +
    done |= !valid + advance + rollout + gather, and valid / penalty against the oracle's restatement of the reference's two functions"""
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS_ALL[name]
    shared = name in SHARED
    if quad is not None:
        if not shared:
            pytest.skip("groups per wave: the shared-phase kernels' option")
        mpk_option("quad", quad)
    eng = engine_of(name)
    T, D = eng.num_steps, eng.num_dof
    B = 1500
    horizon = T
    every_ = every or horizon + 1
    _, closed = specs(name)
    n_ph = int(pc.learn_tau) + int(pc.learn_delay)
    # cfg5: the reference compares action[0], action[1] with the tau / delay bounds although ProMP-TableTennis learns neither
    # (table_tennis_env.py:305-306 on learn_tau = False, mp_wrapper.py:13-18) -- there they are the first two WEIGHTS
    quirk = name == "cfg5_promp_tabletennis"
    chk = n_ph == 2 or quirk
    tb, db = ((0.8, 1.5), (0.05, 0.15)) if quirk else (pc.tau_bound, pc.delay_bound)
    lo = JNT_LOW[:D] * 0.45 if D <= 7 else np.full(D, -1.0)
    hi = JNT_HIGH[:D] * 0.45 if D <= 7 else np.full(D, 1.0)
    gate = dict(pos_low=lo, pos_high=hi, check_tau_delay=chk, tau_bound=tb, delay_bound=db)
    raw0, ip, iv = make_inputs(name, B, seed=3, scale=0.35)
    if quirk:
        rq = np.random.default_rng(17)
        raw0[:, 0] = rq.uniform(0.75, 1.55, B); raw0[:, 1] = rq.uniform(0.04, 0.16, B)
    ip *= 0.3
    lo_hi = O.params_bounds(pc, bc, tc)
    frozen = np.clip(raw0[:, :n_ph], lo_hi[0, :n_ph], lo_hi[1, :n_ph])
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    A, S = _state(B, q0, qd0), _state(B, q0, qd0)
    condA = condS = (cu(ip), cu(iv))
    rng = np.random.default_rng(11)
    segments = O.replanning_segments(horizon, every_, mpt)
    seen_invalid = seen_valid = 0
    for k, (cur, length) in enumerate(segments):
        raw = (0.35 * rng.standard_normal(raw0.shape)).astype(np.float32)
        raw[:, :n_ph] = raw0[:, :n_ph]
        if quirk:
            raw[:, :2] = raw0[:, :2]
        params = raw.copy()
        params[:, :n_ph] = frozen
        it = cur * dt
        g = dict(gate, raw_params=raw)
        if lean:
            r = eng.episode_return(params, condA[0], condA[1], closed, A["q"], A["qd"], replan=(A["ts"], A["ps"], A["dn"], every_, mpt, horizon),
                                   init_time=it, condition=True, gate=g)
            assert ("k_episode_return" in eng.last_kernel()) if shared else ("lean" in eng.last_kernel()), eng.last_kernel()
        else:
            r = eng.replan_step(params, condA[0], condA[1], closed, A["q"], A["qd"], A["ts"], A["ps"], A["dn"], every_, mpt, horizon,
                                init_time=it, condition=True, gate=g)
            assert eng.last_kernel().endswith("closed,gate>") if shared else eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
        s = _separate_step(eng, params, condS[0], condS[1], closed, S["q"], S["qd"], S["ts"], S["ps"], S["dn"], every_, mpt, horizon, it,
                           gate=gate, raw=raw)
        torch.cuda.synchronize()
        valid = r["valid"].cpu().numpy().astype(bool)
        eq(valid, s["valid"].cpu().numpy(), f"plan {k}: valid")
        pen, pen_s = r["penalty"].cpu().numpy(), s["penalty"].cpu().numpy()
        assert np.all(np.abs(pen - pen_s) <= 1e-12 * np.maximum(np.abs(pen_s), 1e-300) + 1e-300), np.abs(pen - pen_s).max()
        for key in ("q", "qd", "ts", "ps", "dn"):
            eq(A[key], S[key], f"plan {k}: {key}")
        eq(r["seg_len"], s["seg_len"], f"plan {k}: seg_len"); eq(r["done"], S["dn"], f"plan {k}: done snapshot")
        eq(r["cond_pos"], s["cond_pos"], f"plan {k}: cond_pos"); eq(r["cond_vel"], s["cond_vel"], f"plan {k}: cond_vel")
        if not lean:
            eq(r["pos"], s["pos"], f"plan {k}: pos"); eq(r["vel"], s["vel"], f"plan {k}: vel"); eq(r["actions"], s["actions"], f"plan {k}: actions")
            bad = ~valid
            assert float(r["actions"][cu(bad)].abs().max() if bad.any() else 0.0) == 0.0
        # the oracle on the GPU's own plan: check_traj_validity + _get_traj_invalid_penalty
        posn = s["pos"].cpu().numpy()
        inside = np.all((posn.astype(np.float64) >= lo) & (posn.astype(np.float64) <= hi), axis=(1, 2))
        if chk:
            a = raw.astype(np.float64)
            inside &= (a[:, 0] >= tb[0]) & (a[:, 0] <= tb[1]) & (a[:, 1] >= db[0]) & (a[:, 1] <= db[1])
        eq(valid, inside, f"plan {k}: valid vs oracle")
        ref_pen = O.traj_invalid_penalty(raw, posn, lo, hi, tb if chk else None, db if chk else None)
        assert np.all(np.abs(pen - ref_pen) <= 1e-12 * np.abs(ref_pen) + 1e-300), np.abs(pen - ref_pen).max()
        live_before = int((r["seg_len"] > 0).sum())
        seen_invalid += int((~valid).sum()); seen_valid += live_before
        condA = (r["cond_pos"], r["cond_vel"]); condS = (s["cond_pos"], s["cond_vel"])
    assert seen_invalid > 20 and seen_valid > 20, (seen_invalid, seen_valid)      # both branches were exercised


@pytest.mark.parametrize("name", ["tt_prodmp", "beerpong_promp", "prodmp_3dof_learn_tau"])
@pytest.mark.parametrize("opt", [("phase_chunk", 1), ("phase_chunk", 2), ("phase_chunk", 4), ("phase_table", 0), ("phase_table", 1),
                                 ("pd_generic", 1), ("write_through", 0), ("write_through", 1), ("tiles_wpb", 1), ("phase_pipe", 0), ("phase_pipe", 1)])
def test_every_launch_geometry_of_the_fused_kernel_gives_the_same_bits(name, opt, mpk_option):
    eng = engine_of(name)
    B = 333
    params, ip, iv = make_inputs(name, B, seed=4)
    _, closed = specs(name)
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    ref = eng.trajectory(params, ip, iv, 0.0)
    q2, qd2 = cu(q0).clone(), cu(qd0).clone()
    act2 = eng.pd_rollout(closed, ref[0], ref[1], q2, qd2)
    mpk_option(*opt)
    q, qd = cu(q0).clone(), cu(qd0).clone()
    pos, vel, act = eng.trajectory_rollout(params, ip, iv, closed, q, qd)
    assert eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
    torch.cuda.synchronize()
    eq(pos, ref[0], "pos"); eq(vel, ref[1], "vel"); eq(act, act2, "actions"); eq(q, q2, "q"); eq(qd, qd2, "qd")


@pytest.mark.parametrize("name", ["tt_prodmp", "beerpong_promp", "promp_5dof_learn_both", "prodmp_3dof_learn_tau", "promp_t49"])
@pytest.mark.parametrize("split", [None, 1, 2, 3, 5, 64])
@pytest.mark.parametrize("B", [2, 333])
def test_the_tiles_of_a_chunk_on_several_waves_give_the_same_bits(name, split, B, mpk_option):
    """frozen-state actions of a small launch: a chunk's 16-step tiles in several wave trips ("phase_split"; automatic below two waves per
    SIMD).  Every split equals the trajectory launch + the rollout on it bit for bit -- also ProMP, whose tiles chain through the position
    carry (re-evaluated at a unit's first step), and a ProMP horizon of 16 n + 1 steps, which stays whole (its last tile reads the velocity
    carry of the tile before it)"""
    if name == "promp_t49":
        CONFIGS_ALL[name] = CONFIGS["promp_5dof_learn_both"][:4] + (0.98,) + CONFIGS["promp_5dof_learn_both"][5:]
    try:
        eng = engine_of(name)
        if name == "promp_t49":
            assert eng.num_steps == 49
        params, ip, iv = make_inputs(name, B, seed=11)
        static, _ = specs(name)
        cp, cv = ip.astype(np.float64) * 0.5, iv.astype(np.float64) * 0.25
        ref = eng.trajectory(params, ip, iv, 0.0)
        act2 = eng.pd_rollout(static, ref[0], ref[1], cu(cp), cu(cv))
        if split is not None:
            mpk_option("phase_split", split)
        pos, vel, act = eng.trajectory_actions(params, ip, iv, static, cp, cv)
        assert eng.last_kernel().startswith("k_phase_fused<") and eng.last_kernel().endswith("act>"), eng.last_kernel()
        torch.cuda.synchronize()
        eq(pos, ref[0], "pos"); eq(vel, ref[1], "vel"); eq(act, act2, "actions")
    finally:
        CONFIGS_ALL.pop("promp_t49", None)


@pytest.mark.parametrize("name", ["tt_prodmp_replan", "beerpong_promp", "promp_5dof_learn_both", "promp_16dof_learn_tau"])
@pytest.mark.parametrize("lean", [False, True])
@pytest.mark.parametrize("B", [1, 5, 1025, 2049, 4096])
def test_the_producer_consumer_form_of_the_closed_loop_gives_the_same_bits(name, lean, B, mpk_option):
    """k_phase_fused<.., pipe> (a consumer wave and three producers per chunk, LDS counters between them; automatic for small launches)
    against the one-wave form on the same inputs: two plans of an episode with the validity gate on, some plans invalid -- every output,
    the plant state, the integer state, the boundary condition, valid and penalty are the same bits; batch sizes around the rule's
    switches between chunks of four and eight and partial last chunks"""
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS[name]
    eng = engine_of(name)
    T, D = eng.num_steps, eng.num_dof
    every_ = every or T // 2 + 3
    mpt_ = max(mpt, 2)
    _, closed = specs(name)
    n_ph = int(pc.learn_tau) + int(pc.learn_delay)
    lo, hi = (JNT_LOW[:D] * 0.45, JNT_HIGH[:D] * 0.45) if D <= 7 else (np.full(D, -0.6), np.full(D, 0.6))
    gate = dict(pos_low=lo, pos_high=hi, check_tau_delay=n_ph == 2, tau_bound=pc.tau_bound if n_ph == 2 else (0.0, 1.0),
                delay_bound=pc.delay_bound if n_ph == 2 else (0.0, 1.0))
    raw0, ip, iv = make_inputs(name, B, seed=21, scale=0.35)
    ip *= 0.3
    lo_hi = O.params_bounds(pc, bc, tc)
    frozen = np.clip(raw0[:, :n_ph], lo_hi[0, :n_ph], lo_hi[1, :n_ph])
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    results = {}
    for mode in (0, 1):
        mpk_option("phase_pipe", mode)
        A = _state(B, q0, qd0)
        cond = (cu(ip), cu(iv))
        rng = np.random.default_rng(5)
        outs = []
        for k, (cur, length) in enumerate(O.replanning_segments(T, every_, mpt_)[:2]):
            raw = (0.35 * rng.standard_normal(raw0.shape)).astype(np.float32)
            raw[:, :n_ph] = raw0[:, :n_ph]
            params = raw.copy()
            params[:, :n_ph] = frozen
            g = dict(gate, raw_params=raw)
            if lean:
                r = eng.episode_return(params, cond[0], cond[1], closed, A["q"], A["qd"], replan=(A["ts"], A["ps"], A["dn"], every_, mpt_, T),
                                       init_time=cur * dt, condition=True, gate=g)
            else:
                r = eng.replan_step(params, cond[0], cond[1], closed, A["q"], A["qd"], A["ts"], A["ps"], A["dn"], every_, mpt_, T,
                                    init_time=cur * dt, condition=True, gate=g)
            assert ("pipe" in eng.last_kernel()) == bool(mode), eng.last_kernel()
            torch.cuda.synchronize()
            eng.check_range()
            outs.append({key: r[key].clone() for key in r if torch.is_tensor(r[key])})
            outs[-1].update({key: A[key].clone() for key in A})
            cond = (r["cond_pos"], r["cond_vel"])
        results[mode] = outs
    for k, (one, pipe) in enumerate(zip(results[0], results[1])):
        assert one.keys() == pipe.keys()
        for key in one:
            eq(pipe[key], one[key], f"plan {k}: {key}")
    if B >= 1000:
        v = results[1][0]["valid"].cpu().numpy().astype(bool)
        assert v.any() and (~v).any()           # both branches


@pytest.mark.parametrize("name", ["tt_prodmp", "beerpong_promp"])
@pytest.mark.parametrize("ctrl", ["velocity", "position"])
def test_the_other_controllers_and_finite_action_bounds(name, ctrl):
    pc, bc, tc, dt, dur, (pg, dg) = CONFIGS[name][:6]
    eng = engine_of(name)
    B = 130
    params, ip, iv = make_inputs(name, B, seed=6)
    low, high = np.linspace(-0.7, -0.2, 7), np.linspace(0.1, 0.9, 7)
    static, closed = specs(name, low, high, ctrl)
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    q, qd = cu(q0).clone(), cu(qd0).clone()
    pos, vel, act = eng.trajectory_rollout(params, ip, iv, closed, q, qd)
    assert eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
    torch.cuda.synchronize()
    ra, rq, rqd = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), ctrl, pg, dg, low, high, "double_integrator", dt, q0, qd0)
    eq(act, ra.astype(np.float32), "actions"); eq(q, rq, "q"); eq(qd, rqd, "qd")
    pos, vel, act = eng.trajectory_actions(params, ip, iv, static, q0, qd0)
    torch.cuda.synchronize()
    ra, _, _ = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), ctrl, pg, dg, low, high, "static", dt, q0, qd0)
    eq(act, ra.astype(np.float32), "static actions")


@pytest.mark.parametrize("name", ["tt_prodmp", "beerpong_promp"])
def test_output_arrays_that_are_not_16_byte_aligned(name):
    eng = engine_of(name)
    T, D = eng.num_steps, eng.num_dof
    B = 37
    params, ip, iv = make_inputs(name, B, seed=7)
    _, closed = specs(name)
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    ref = eng.trajectory_rollout(params, ip, iv, closed, cu(q0).clone(), cu(qd0).clone())
    for off in (1, 2, 3):
        bufs = [torch.full((B * T * D + 8,), 7.0, device="cuda") for _ in range(3)]
        out = tuple(b[off:off + B * T * D].view(B, T, D) for b in bufs)
        got = eng.trajectory_rollout(params, ip, iv, closed, cu(q0).clone(), cu(qd0).clone(), out=out)
        torch.cuda.synchronize()
        for g, r, w in zip(got, ref, ("pos", "vel", "actions")):
            eq(g, r, f"{w} at offset {off}")
        for b in bufs:      # nothing outside the arrays was touched
            assert float(b[:off].min()) == 7.0 and float(b[off + B * T * D:].min()) == 7.0 and float(b[off + B * T * D:].max()) == 7.0


def test_the_golden_fixture_of_tabletennis_prodmp_through_the_fused_entry_points():
    """tests/golden/tt_prodmp_learn_tau_delay.npz (the oracle's and the second formulation's outputs for the TableTennis-ProDMP
    configuration, plus the reference loop's actions / final state on the oracle's fp32 plan): the fused launches reproduce the
    fixture's trajectories to 1e-5; actions follow the plan they were computed from (compared where the plans agree bit for bit)"""
    from tests.golden.make_golden import CONFIGS as GOLD
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tt_prodmp_learn_tau_delay.npz"), allow_pickle=False)
    cfg = GOLD["tt_prodmp_learn_tau_delay"]
    pc, bc, tc, dt, dur = cfg["pc"], cfg["bc"], cfg["tc"], cfg["dt"], cfg["duration"]
    pg, dg = cfg["gains"]
    lo, hi = cfg["act"]
    eng = make_engine(pc, bc, tc, dt, dur, device=0)
    params, ip, iv = z["params"], z["init_pos"], z["init_vel"]
    static = RolloutSpec("motor", tc.action_dim, pg, dg, lo, hi, plant="static")
    closed = RolloutSpec("motor", tc.action_dim, pg, dg, lo, hi, plant="double_integrator", dt=dt)
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    pos, vel, act = eng.trajectory_actions(params, ip, iv, static, q0, qd0, float(z["init_times"][0]))
    assert eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
    q, qd = cu(q0).clone(), cu(qd0).clone()
    pos2, vel2, act2 = eng.trajectory_rollout(params, ip, iv, closed, q, qd, init_time=float(z["init_times"][0]))
    assert eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
    torch.cuda.synchronize()
    for ref_p, ref_v in ((z["pos32_0"], z["vel32_0"]), (z["tpos_0"], z["tvel_0"])):
        close(pos.cpu().numpy(), ref_p, "pos vs fixture"); close(vel.cpu().numpy(), ref_v, "vel vs fixture")
        close(pos2.cpu().numpy(), ref_p, "pos (closed) vs fixture"); close(vel2.cpu().numpy(), ref_v, "vel (closed) vs fixture")
    # actions are clip(p (des - q) + d (desv - qd)): 1e-5 on the plan scale moves them by at most max gain x that
    scale = max(np.abs(z["pos32_0"]).max(), np.abs(z["vel32_0"]).max())
    assert np.abs(act.cpu().numpy() - z["act_static"]).max() <= 4.0 * 1e-5 * scale * float(np.max(pg) + np.max(dg))


@pytest.mark.parametrize("name", REFERENCE)
def test_the_reference_shapes_at_8192_episodes(name):
    """the size the review asked for: fused == separate launches bit for bit at B = 8192, oracle on a sample"""
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS[name]
    eng = engine_of(name)
    B = 8192
    T = eng.num_steps
    params, ip, iv = make_inputs(name, B, seed=8, scale=0.5)
    _, closed = specs(name)
    q0, qd0 = ip.astype(np.float64), iv.astype(np.float64)
    A, S = _state(B, q0, qd0), _state(B, q0, qd0)
    every_ = every or T + 1
    r = eng.replan_step(params, ip, iv, closed, A["q"], A["qd"], A["ts"], A["ps"], A["dn"], every_, mpt, T, condition=True)
    assert eng.last_kernel().startswith("k_phase_fused<"), eng.last_kernel()
    s = _separate_step(eng, params, cu(ip), cu(iv), closed, S["q"], S["qd"], S["ts"], S["ps"], S["dn"], every_, mpt, T, 0.0)
    torch.cuda.synchronize()
    for key in ("pos", "vel", "actions", "seg_len", "cond_pos", "cond_vel"):
        eq(r[key], s[key], key)
    for key in ("q", "qd", "ts", "ps", "dn"):
        eq(A[key], S[key], key)
    idx = np.random.default_rng(0).choice(B, 256, replace=False)
    rp, rv = O.get_trajectory(pc, bc, tc, params[idx], dur, dt, 0.0, ip[idx], iv[idx], dtype=np.float64)
    close(r["pos"][cu(idx)].cpu().numpy(), rp, "pos sample")
    close(r["vel"][cu(idx)].cpu().numpy(), rv, "vel sample", atol=fd_atol(rp, dt) if tc.trajectory_generator_type == "promp" else 0.0)
    seg = r["seg_len"].cpu().numpy()[idx]
    ra, rq, rqd = O.rollout(r["pos"][cu(idx)].cpu().numpy(), r["vel"][cu(idx)].cpu().numpy(), "motor", pg, dg, -1.0, 1.0, "double_integrator",
                            dt, q0[idx], qd0[idx], n_steps=seg)
    eq(r["actions"][cu(idx)], ra.astype(np.float32), "actions sample"); eq(A["q"][cu(idx)], rq, "q sample"); eq(A["qd"][cu(idx)], rqd, "qd sample")


# ---- BatchedBlackBox on the reference's learned-phase families ---------------------------------------------------------------------
def _bb(name, B, **kw):
    from fancy_gym_amd import BatchedBlackBox
    from fancy_gym_amd.black_box.factory import get_basis_generator, get_controller, get_phase_generator, get_trajectory_generator
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS_ALL[name]
    phase = get_phase_generator(pc.phase_generator_type, tau=pc.tau, alpha_phase=pc.alpha_phase, learn_tau=pc.learn_tau,
                                learn_delay=pc.learn_delay, tau_bound=list(pc.tau_bound), delay_bound=list(pc.delay_bound))
    bkw = dict(num_basis=bc.num_basis, basis_bandwidth_factor=bc.basis_bandwidth_factor)
    if bc.basis_generator_type == "prodmp":
        bkw["alpha"] = bc.alpha
    if bc.basis_generator_type == "zero_rbf":
        bkw.update(num_basis_zero_start=bc.num_basis_zero_start, num_basis_zero_goal=bc.num_basis_zero_goal)
    basis = get_basis_generator(bc.basis_generator_type, phase, **bkw)
    tkw = dict(weights_scale=tc.weights_scale)
    if tc.trajectory_generator_type == "prodmp":
        tkw.update(goal_scale=tc.goal_scale, auto_scale_basis=tc.auto_scale_basis, disable_goal=tc.disable_goal, relative_goal=tc.relative_goal)
    tg = get_trajectory_generator(tc.trajectory_generator_type, tc.action_dim, basis, **tkw)
    ctrl = get_controller("motor", p_gains=pg, d_gains=dg)
    if every is not None:
        kw.setdefault("replanning_every", every); kw.setdefault("max_planning_times", mpt)
    return BatchedBlackBox(tg, ctrl, B, dt, dur, act_low=-1.0, act_high=1.0, plant="double_integrator", **kw)


@pytest.mark.parametrize("name", ["tt_prodmp", "tt_prodmp_replan", "beerpong_promp", "cfg5_promp_tabletennis"])
@pytest.mark.parametrize("gated", [False, True])
def test_batched_black_box_steps_these_families_in_one_launch_per_plan(name, gated):
    """BatchedBlackBox.step on the reference's TableTennis / BeerPong configurations: ONE launch per plan (k_phase_fused for the learned
    phase, the lane-quarter kernel for cfg5), with the validity gate inside it, at verbose 2 and 1 -- equal to the separate launches
    (`fuse=False`) plan for plan: trajectories / actions / plant state / counters / flags bit for bit on the first plan (the separate
    launches leave lockstep at the first invalid plan: per-episode times, another kernel family, 2e-6 of the scale from there on)"""
    pc, bc, tc, dt, dur, (pg, dg), every, mpt = CONFIGS_ALL[name]
    B = 700
    D = tc.action_dim
    kw = {}
    if gated:
        kw = dict(pos_limits=(JNT_LOW[:D] * 0.7, JNT_HIGH[:D] * 0.7), check_tau_delay=(int(pc.learn_tau) + int(pc.learn_delay) == 2))
    fused, lean, apart = _bb(name, B, **kw), _bb(name, B, verbose=1, **kw), _bb(name, B, **kw)
    raw0, ip, iv = make_inputs(name, B, seed=12, scale=0.35)
    ip *= 0.3
    n_ph = int(pc.learn_tau) + int(pc.learn_delay)
    for bb in (fused, lean, apart):
        bb.reset(ip.astype(np.float64), iv.astype(np.float64))
    rng = np.random.default_rng(13)
    n_plans = len(O.replanning_segments(fused.horizon, every or fused.horizon + 1, mpt))
    for k in range(n_plans):
        raw = (0.35 * rng.standard_normal(raw0.shape)).astype(np.float32)
        raw[:, :n_ph] = raw0[:, :n_ph] if k == 0 else rng.uniform(0.0, 2.0, (B, n_ph)).astype(np.float32)    # (later plans: ignored, frozen)
        a = fused.step(raw)
        assert fused.engine.last_kernel().startswith("k_phase_fused<") or "closed" in fused.engine.last_kernel(), fused.engine.last_kernel()
        if gated:
            assert "gate" in fused.engine.last_kernel() or fused.engine.last_kernel().startswith("k_phase_fused<"), fused.engine.last_kernel()
        b = lean.step(raw)
        assert "lean" in lean.engine.last_kernel() or "k_episode_return" in lean.engine.last_kernel(), lean.engine.last_kernel()
        c = apart.step(raw, fuse=False)
        torch.cuda.synchronize()
        for key in ("trajectory_length", "done", "valid", "terminated", "truncated"):
            eq(a[key], b[key], f"plan {k}: {key} (verbose 2 / 1)")
            if apart._lockstep is not None or not gated:
                eq(a[key], c[key], f"plan {k}: {key}")
            elif key != "valid":
                # (per-episode times since an invalid plan: 2e-6 of the scale may flip the verdict of a plan that touches a limit; the
                # verdict on a FINISHED episode's plan -- evaluated at the batch's shared time here, at its own frozen time there -- is
                # nobody's business: `valid` is compared where the plan is executed, below)
                assert float((a[key] != c[key]).float().mean()) <= 0.02, (k, key)
        eq(fused.q, lean.q, f"plan {k}: q fused / verbose 1"); eq(fused.qd, lean.qd, f"plan {k}: qd"); eq(fused.traj_steps, lean.traj_steps, "traj_steps")
        assert "des_pos" not in b and "step_actions" not in b
        if gated:
            # (float64 excess sums: k_traj_pipe<.., gate> / k_phase_fused add per lane in step order, k_episode_return per row tile -- the
            # same numbers to the last bits: 1e-12 relative, as against mpk_traj_validity_penalty; include/mpk.h, mpk_replan_step_gated)
            pa, pb = a["invalid_penalty"].cpu().numpy(), b["invalid_penalty"].cpu().numpy()
            assert np.all(np.abs(pa - pb) <= 1e-12 * np.abs(pb) + 1e-300), f"plan {k}: penalty fused / verbose 1: {np.abs(pa - pb).max():.3e}"
        exact = k == 0 or not gated or bool(c["valid"].all())
        if exact and apart._lockstep is not None:
            for key in ("des_pos", "des_vel", "step_actions"):
                eq(a[key], c[key], f"plan {k}: {key}")
            eq(fused.q, apart.q, f"plan {k}: q"); eq(fused.qd, apart.qd, f"plan {k}: qd")
            if gated:
                pa, pb = a["invalid_penalty"].cpu().numpy(), c["invalid_penalty"].cpu().numpy()
                assert np.all(np.abs(pa - pb) <= 1e-12 * np.abs(pb) + 1e-300)
        else:
            live = (a["trajectory_length"] > 0) & (c["trajectory_length"] > 0)
            assert bool(a["valid"][live].all()) and bool(c["valid"][live].all())
            for key in ("des_pos", "des_vel"):
                assert not bool(live.any()) or float((a[key] - c[key])[live].abs().max()) <= 1e-5 * float(c[key].abs().max()), (k, key)
    assert bool(fused.done.all()) and bool(lean.done.all())
    if gated:
        assert int((fused.traj_steps < fused.horizon).sum()) > 0        # some episodes ended at an invalid plan


def test_the_step_flags_of_a_gated_plan_in_one_launch():
    """mpk_gate_flags: terminated = !valid & !was_done, truncated = done & valid (black_box_wrapper.py:169-172,198-203) -- what
    BatchedBlackBox computed with five elementwise launches per gated step; was_done = None right after a reset"""
    eng = engine_of("tt_prodmp")
    g = torch.Generator().manual_seed(3)
    for B in (1, 255, 256, 257, 5000):
        v, w, d = (torch.randint(0, 2, (B,), generator=g, dtype=torch.uint8).cuda() for _ in range(3))
        term, trunc = eng.gate_flags(v, w, d)
        eq(term, ~v.bool() & ~w.bool(), "terminated"); eq(trunc, d.bool() & v.bool(), "truncated")
        term, trunc = eng.gate_flags(v.bool(), None, d.bool())
        eq(term, ~v.bool(), "terminated after a reset"); eq(trunc, d.bool() & v.bool(), "truncated after a reset")
