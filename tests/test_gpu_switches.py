"""
GPU suite: the SURVEY Appendix A "(?)" items as explicit switches (include/mpk.h MPK_RELGOAL_*, MPK_GOAL_OFFSET_*,
MPK_SINGLE_RBF_*, MPK_DMP_FIRST_*) -- BOTH settings of every switch, through every kernel family that implements it,
against the oracle's matching setting.  The defaults are the behaviour every BASELINE test uses.
"""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import mp_oracle as O
from tests.test_gpu_trajectory import CFG3, close, inputs, make_engine

pytestmark = pytest.mark.gpu

# the reference's TableTennis / BoxPushing ProDMP kwargs (table_tennis/mp_wrapper.py:49-54, box_pushing/mp_wrapper.py:72-79)
# with goal_scale != 1 so that the two relative-goal orderings differ visibly
PC = O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0)
BC = O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=3, alpha=10)
TC = O.TrajCfg("prodmp", action_dim=7, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True, relative_goal=True)

# (name, options, per-episode init_time?) -> which kernel family runs
PATHS = [("tiles", {"mapping": 1}, False), ("stream", {"mapping": 2}, False), ("phase", {}, True),
         ("rows", {"phase": 0}, True)]


def _run(pc, bc, tc, dt, dur, B, init_time, per_episode, seed=0):
    eng = make_engine(pc, bc, tc, dt, dur)
    params, ip, iv = inputs(pc, bc, tc, B, seed=seed)
    it = torch.full((B,), init_time, dtype=torch.float32, device="cuda") if per_episode else init_time
    pos, vel = eng.trajectory(params, ip, iv, it)
    torch.cuda.synchronize()
    rp, rv = O.get_trajectory(pc, bc, tc, params, dur, dt, np.full(B, init_time, np.float32) if per_episode else init_time,
                              ip, iv, dtype=np.float64)
    return eng, pos.cpu().numpy(), vel.cpu().numpy(), rp, rv, (params, ip, iv)


@pytest.mark.parametrize("path,opts,per_episode", PATHS, ids=[p[0] for p in PATHS])
@pytest.mark.parametrize("disable_goal", [False, True])
@pytest.mark.parametrize("mode", ["after_scale", "before_scale"])
def test_relative_goal_mode(mode, disable_goal, path, opts, per_episode, mpk_option):
    for k, v in opts.items():
        mpk_option(k, v)
    tc = dataclasses.replace(TC, relative_goal_mode=mode, disable_goal=disable_goal)
    for init_time in (0.0, 0.5):
        eng, pos, vel, rp, rv, _ = _run(PC, BC, tc, 0.02, 2.0, 9, init_time, per_episode, seed=3)
        close(pos, rp, "pos"); close(vel, rv, "vel")
        assert path in eng.last_kernel() or (path == "tiles" and "k_traj_tiles" in eng.last_kernel()), eng.last_kernel()


def test_relative_goal_modes_differ_by_one_minus_scale_times_init_pos():
    """after_scale: goal = s_g*g + y_b; before_scale: goal = s_g*(g + y_b).  The trajectories end (s -> 6 tau) at the goal"""
    outs = {}
    for mode in ("after_scale", "before_scale"):
        tc = dataclasses.replace(TC, relative_goal_mode=mode)
        pc = O.PhaseCfg("exp", tau=0.4, alpha_phase=3.0)      # 2 s = 5 tau: the attractor has converged
        _, pos, _, _, _, (params, ip, iv) = _run(pc, BC, tc, 0.02, 2.0, 5, 0.0, False, seed=1)
        outs[mode] = pos
    eng = make_engine(PC, BC, TC, 0.02, 2.0)
    s_g = np.float32(eng.prodmp_tables()["scale"][5]) * np.float32(0.3)
    diff = outs["after_scale"][:, -1] - outs["before_scale"][:, -1]
    np.testing.assert_allclose(diff, (1.0 - s_g) * ip, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("path,opts,per_episode", PATHS, ids=[p[0] for p in PATHS])
@pytest.mark.parametrize("relative,disable_goal", [(False, False), (True, False), (False, True)])
def test_goal_offset_add(relative, disable_goal, path, opts, per_episode, mpk_option):
    for k, v in opts.items():
        mpk_option(k, v)
    tc = dataclasses.replace(TC, relative_goal=relative, disable_goal=disable_goal, goal_offset_mode="add",
                             goal_offset=1.0)
    for init_time in (0.0, 0.5):
        eng, pos, vel, rp, rv, _ = _run(PC, BC, tc, 0.02, 2.0, 9, init_time, per_episode, seed=5)
        close(pos, rp, "pos"); close(vel, rv, "vel")
    if not per_episode:
        assert eng.last_kernel().startswith("k_traj_"), eng.last_kernel()


def test_goal_offset_ignore_is_bitwise_the_configuration_without_an_offset():
    """default: the kwarg is accepted and dropped (box_pushing/mp_wrapper.py:77 passes goal_offset = 1.0)"""
    tc_off = dataclasses.replace(TC, goal_offset_mode="ignore", goal_offset=1.0)
    a = _run(PC, BC, tc_off, 0.02, 2.0, 17, 0.25, False, seed=2)
    b = _run(PC, BC, TC, 0.02, 2.0, 17, 0.25, False, seed=2)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # an offset of zero in 'add' mode needs no extra contraction column either
    tc_zero = dataclasses.replace(TC, goal_offset_mode="add", goal_offset=0.0)
    c = _run(PC, BC, tc_zero, 0.02, 2.0, 17, 0.25, False, seed=2)
    assert np.array_equal(c[1], b[1])


@pytest.mark.parametrize("quad", ["0", "2", "3", "4"])
def test_goal_offset_through_the_fused_closed_loop_kernels(quad, mpk_option):
    """an extra contraction column (KM = 3) through k_traj_stream / quad / duo / mono<closed> and the open-loop actions"""
    from fancy_gym_amd import RolloutSpec
    mpk_option("quad", quad)
    tc = dataclasses.replace(TC, goal_offset_mode="add", goal_offset=-0.7)
    eng = make_engine(PC, BC, tc, 0.02, 2.0)
    B, D = 37, 7
    params, ip, iv = inputs(PC, BC, tc, B, seed=9)
    pg, dg = np.linspace(0.5, 1.5, D), np.linspace(0.05, 0.15, D)
    rp, rv = O.get_trajectory(PC, BC, tc, params, 2.0, 0.02, 0.0, ip, iv, dtype=np.float64)
    spec = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="double_integrator", dt=0.02)
    q = torch.tensor(ip, dtype=torch.float64, device="cuda")
    qd = torch.zeros_like(q)
    pos, vel, act = eng.trajectory_rollout(params, ip, iv, spec, q, qd)
    torch.cuda.synchronize()
    close(pos.cpu().numpy(), rp, "pos"); close(vel.cpu().numpy(), rv, "vel")
    ra, rq, rqd = O.rollout(pos.cpu().numpy(), vel.cpu().numpy(), "motor", pg, dg, -1.0, 1.0, "double_integrator", 0.02,
                            ip.astype(np.float64), np.zeros((B, D)))
    assert np.array_equal(act.cpu().numpy(), ra.astype(np.float32))
    assert np.array_equal(q.cpu().numpy(), rq) and np.array_equal(qd.cpu().numpy(), rqd)
    spec_s = RolloutSpec("motor", D, pg, dg, -1.0, 1.0, plant="static")
    p2, v2, a2 = eng.trajectory_actions(params, ip, iv, spec_s, ip.astype(np.float64), iv.astype(np.float64))
    torch.cuda.synchronize()
    assert torch.equal(p2, pos) and torch.equal(v2, vel)
    ra2, _, _ = O.rollout(p2.cpu().numpy(), v2.cpu().numpy(), "motor", pg, dg, -1.0, 1.0, "static", 0.02,
                          ip.astype(np.float64), iv.astype(np.float64))
    assert np.array_equal(a2.cpu().numpy(), ra2.astype(np.float32))


@pytest.mark.parametrize("basis,kw", [("rbf", {}), ("zero_rbf", dict(num_basis_zero_start=0, num_basis_zero_goal=0))])
def test_single_rbf_mode(basis, kw):
    pc = O.PhaseCfg("linear", tau=2.0)
    for mode in ("unit_gap", "refuse"):
        bc = O.BasisCfg(basis, num_basis=1, single_rbf_mode=mode, **kw)
        tc = O.TrajCfg("promp", action_dim=3)
        if mode == "refuse":
            with pytest.raises(ValueError, match="single radial basis"):
                make_engine(pc, bc, tc, 0.02, 2.0)
            with pytest.raises(ValueError, match="single radial basis"):
                O.rbf_centers_bandwidth(pc, bc)
        else:
            eng, pos, vel, rp, rv, _ = _run(pc, bc, tc, 0.02, 2.0, 5, 0.0, False)
            close(pos, rp, "pos")
    # two basis functions in total are never refused
    bc2 = O.BasisCfg("zero_rbf", num_basis=1, num_basis_zero_start=1, num_basis_zero_goal=0, single_rbf_mode="refuse")
    make_engine(pc, bc2, O.TrajCfg("promp", action_dim=3), 0.02, 2.0)


DMP_PATHS = [("quad4", {"quad": 2}, False, False), ("duo", {"quad": 3}, False, False), ("stream", {"quad": 0}, False, False),
             ("bulk", {"quad": 0, "bulk": 2}, False, False), ("phase", {}, True, False), ("rows", {"phase": 0}, True, False),
             ("learn_tau", {}, False, True)]


@pytest.mark.parametrize("path,opts,per_episode,learn", DMP_PATHS, ids=[p[0] for p in DMP_PATHS])
@pytest.mark.parametrize("mode", ["init", "step"])
def test_dmp_first_sample(mode, path, opts, per_episode, learn, mpk_option):
    for k, v in opts.items():
        mpk_option(k, v)
    pc, bc, tc, dt, dur = CFG3
    if learn:
        pc = dataclasses.replace(pc, learn_tau=True, learn_delay=True, tau_bound=(2.0, 4.0), delay_bound=(0.0, 0.3))
    tc = dataclasses.replace(tc, dmp_first_sample=mode)
    for init_time, B in ((0.0, 33), (0.3, 5)):
        eng, pos, vel, rp, rv, (params, ip, iv) = _run(pc, bc, tc, dt, dur, B, init_time, per_episode, seed=7)
        close(pos, rp, "pos", rtol=2e-5); close(vel, rv, "vel", rtol=2e-5)
        if mode == "init":
            assert np.array_equal(pos[:, 0], ip)
        else:
            assert not np.array_equal(pos[:, 0], ip)


def test_dmp_first_sample_step_equals_init_mode_started_one_step_later():
    """'step' = one Euler step from the boundary state, then the ordinary recurrence: feeding 'init' mode the stepped state
    (the oracle's first 'step' sample) reproduces the 'step' trajectory"""
    pc, bc, tc, dt, dur = CFG3
    B = 12
    params, ip, iv = inputs(pc, bc, tc, B, seed=4)
    e_step = make_engine(pc, bc, dataclasses.replace(tc, dmp_first_sample="step"), dt, dur)
    e_init = make_engine(pc, bc, tc, dt, dur)
    p1, v1 = e_step.trajectory(params, ip, iv, 0.0)
    p0, v0 = e_init.trajectory(params, p1[:, 0].contiguous(), v1[:, 0].contiguous(), 0.0)
    torch.cuda.synchronize()
    close(p0.cpu().numpy(), p1.cpu().numpy(), "pos", rtol=2e-6)
    close(v0.cpu().numpy(), v1.cpu().numpy(), "vel", rtol=2e-6)
