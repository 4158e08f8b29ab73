"""
GPU suite: the HIP path against an INDEPENDENT statement of what a movement primitive is -- SciPy's solution of the
DMP / ProDMP differential equation (tests/ode_ref.py) -- instead of against the oracle's recollection of mp_pytorch.
Covers BASELINE cfg2, cfg4 (all four replanning boundary conditions, auto_scale_basis, disable_goal), the reference's
TableTennis ProDMP configuration with learned tau / delay (auto_scale_basis, relative_goal, disable_goal), and cfg3
(explicit Euler: first-order convergence to the ODE in dt).

Tolerances are derived from the ProDMP table step h = basis_dt / tau0 (scaled time):
  * against the ODE evaluated AT the table's grid points (s = idx * h, boundary at idx_b * h) the only error is the
    cumulative-trapezoid quadrature of the pre-computed integrals, O(h^2): measured 3e-6 .. 2e-5 of the trajectory scale for
    the configurations here; bound used: 40 * h^2 * scale (h = 1/150: 1.8e-3 ... the constant covers alpha = 25);
  * against the ODE at the EXACT sample times the table lookup adds |dy/ds| * h / 2 (index rounding, half a grid step).
"""
import numpy as np
import pytest
import torch

from tests import ode_ref as R
from tests.test_gpu_trajectory import CFG2, CFG3, CFG4, make_engine
from oracle import mp_oracle as O

pytestmark = pytest.mark.gpu


def _prodmp_case(pc, bc, tc, dt, duration, params, ip, iv, init_time, tau=None, delay=None, factory=None):
    """returns (pos, vel) from the GPU and the ODE solution at the table grid points, plus the table step"""
    eng = (factory or make_engine)(pc, bc, tc, dt, duration)
    B, D = ip.shape
    it_arg = init_time
    if isinstance(init_time, np.ndarray):
        it_arg = torch.tensor(init_time.astype(np.float32), device=eng.device)
    pos, vel = eng.trajectory(params, ip, iv, it_arg)
    if pc.learn_tau or pc.learn_delay:
        eng.check_range()                   # synchronises
    pos, vel = pos.cpu().numpy().astype(np.float64), vel.cpu().numpy().astype(np.float64)
    tabs = eng.prodmp_tables()
    cen, bw = R.rbf_constants(eng)
    nb = bc.num_basis
    scale = np.ones(nb + 1)
    if tc.auto_scale_basis:
        scale = tabs["scale"].astype(np.float32).astype(np.float64)
    scale[:nb] *= np.float32(tc.weights_scale); scale[nb] *= np.float32(tc.goal_scale)
    h = float(np.float32(bc.dt) / np.float32(pc.tau))
    off = int(pc.learn_tau) + int(pc.learn_delay)
    local = params[:, off:].reshape(B, D, -1).astype(np.float64)
    w = np.zeros((B, D, nb)); g = np.zeros((B, D))
    c = 0
    if not tc.disable_weights:
        w = local[..., :nb] * scale[:nb]; c = nb
    if not tc.disable_goal:
        g = local[..., c] * scale[nb]
    if tc.relative_goal:                       # default mode: init_pos joins the RAW goal parameter, the scale applies to both
        g = g + ip.astype(np.float64) * (scale[nb] if tc.relative_goal_mode == "before_scale" else 1.0)
    base = eng.times()
    out_y, out_v = np.empty_like(pos), np.empty_like(vel)
    slope = np.zeros(B)
    for b in range(B):
        tau_b = np.float32(pc.tau if tau is None else tau[b])
        delay_b = np.float32(pc.delay if delay is None else delay[b])
        it_b = np.float32(init_time[b] if isinstance(init_time, np.ndarray) else init_time)
        idx = R.table_indices(base + it_b, tau_b, delay_b, np.float32(h))
        idx_b = R.table_indices(np.array([it_b], np.float32), tau_b, delay_b, np.float32(h))[0]
        y, ys = R.solve(float(np.float32(bc.alpha)), float(np.float32(pc.alpha_phase)), cen, bw, w[b], g[b],
                        idx_b * h, ip[b].astype(np.float64), iv[b].astype(np.float64) * float(tau_b), idx * h)
        out_y[b] = y.T
        out_v[b] = ys.T / float(tau_b)
        slope[b] = np.abs(ys).max()
    return pos, vel, out_y, out_v, h, slope


def _check(pos, vel, y, v, h, alpha):
    sp = max(np.abs(y).max(), 1e-3)
    sv = max(np.abs(v).max(), 1e-3)
    # O(h^2) trapezoid residual of the pre-computed integrals (+ the fp32 contract of the kernels, 1e-5)
    tol_p = (40.0 * h * h + 1e-5) * sp
    tol_v = (40.0 * h * h * alpha / 4 + 1e-5) * sv       # the velocity rows carry an extra factor ~alpha/2 of curvature
    ep, ev = np.abs(pos - y).max(), np.abs(vel - v).max()
    assert ep <= tol_p, f"pos: {ep:.3e} > {tol_p:.3e} (scale {sp:.3e})"
    assert ev <= tol_v, f"vel: {ev:.3e} > {tol_v:.3e} (scale {sv:.3e})"
    return ep / sp, ev / sv


def _inputs(P, D, B, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, P)).astype(np.float32), rng.uniform(-1, 1, (B, D)).astype(np.float32),
            rng.uniform(-1, 1, (B, D)).astype(np.float32))


def check_cfg2(init_time, factory=None):
    pc, bc, tc, dt, dur = CFG2
    params, ip, iv = _inputs(42, 7, 6, 11)
    pos, vel, y, v, h, _ = _prodmp_case(pc, bc, tc, dt, dur, params, ip, iv, init_time, factory=factory)
    _check(pos, vel, y, v, h, bc.alpha)


@pytest.mark.parametrize("mapping", ["1", "2"])
@pytest.mark.parametrize("init_time", [0.0, 0.5])
def test_cfg2_prodmp_solves_the_ode(init_time, mapping, mpk_option):
    mpk_option("mapping", mapping)
    check_cfg2(init_time)


def test_cfg4_all_four_replanning_boundary_conditions_solve_the_ode():
    check_cfg4()


def check_cfg4(factory=None):
    """
    BoxPushingDenseReplan (box_pushing/mp_wrapper.py:68-92): weights_scale = goal_scale = 0.3, auto_scale_basis,
    disable_goal, plans at t = 0, 25, 50, 75 steps, each conditioned on the DESIRED state where the previous plan broke
    (condition_on_desired, black_box_wrapper.py:199-201).  Every plan must solve the ODE from its own boundary state;
    with unchanged parameters consecutive plans continue ONE solution, so plan k must also reproduce plan 0's tail.
    """
    pc, bc, tc, dt, dur = CFG4
    B, D = 5, 7
    rng = np.random.default_rng(4)
    prm_same = rng.standard_normal((B, 35)).astype(np.float32)
    ip = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    iv = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    first = None
    for same in (True, False):
        cp, cv = ip, iv
        for k in range(4):
            prm = prm_same if same else rng.standard_normal((B, 35)).astype(np.float32)
            init_time = float(np.float32(25 * k * dt))
            pos, vel, y, v, h, _ = _prodmp_case(pc, bc, tc, dt, dur, prm, cp, cv, init_time, factory=factory)
            _check(pos, vel, y, v, h, bc.alpha)
            if same:
                if k == 0:
                    first = (pos, vel)
                else:
                    # plan k from the desired state at step 25k - 1 of plan k-1 == the tail of plan 0.  The boundary
                    # state is an fp32 sample of plan k-1, so the continuation is exact up to that rounding carried
                    # through the (stable) dynamics
                    n = 100 - 25 * k
                    sp, sv = np.abs(first[0]).max(), np.abs(first[1]).max()
                    assert np.abs(pos[:, :n] - first[0][:, 25 * k:]).max() <= 2e-5 * sp
                    assert np.abs(vel[:, :n] - first[1][:, 25 * k:]).max() <= 2e-5 * sv * bc.alpha
            # condition_on_desired: the next plan starts from the desired state at the last executed step
            cp, cv = pos[:, 24].astype(np.float32), vel[:, 24].astype(np.float32)


@pytest.mark.parametrize("phase_opt", ["1", "0"])
def test_tabletennis_prodmp_with_learned_tau_and_delay_solves_the_ode(phase_opt, mpk_option):
    mpk_option("phase", phase_opt)
    check_tabletennis_prodmp()


def check_tabletennis_prodmp(factory=None):
    """
    table_tennis/mp_wrapper.py:33-55: learn_tau, learn_delay, tau_bound [0.8, 1.5], delay_bound [0.05, 0.15],
    alpha_phase 3, 3 basis functions, alpha 25, weights_scale 0.7, auto_scale_basis, relative_goal, disable_goal;
    dt = 0.008, 350 steps.  Per-episode phase -> the per-episode kernels (wave-per-episode and workgroup-per-episode).
    """
    pc = O.PhaseCfg("exp", tau=2.8, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5),
                    delay_bound=(0.05, 0.15))
    bc = O.BasisCfg("prodmp", num_basis=3, basis_bandwidth_factor=3, alpha=25)
    tc = O.TrajCfg("prodmp", action_dim=7, weights_scale=0.7, auto_scale_basis=True, relative_goal=True,
                   disable_goal=True)
    B, D = 6, 7
    rng = np.random.default_rng(8)
    params = rng.standard_normal((B, 2 + 21)).astype(np.float32)
    params[:, 0] = rng.uniform(0.8, 1.5, B)
    params[:, 1] = rng.uniform(0.05, 0.15, B)
    params[0, 0], params[1, 0] = 0.3, 9.0            # outside the bounds: clipped to 0.8 / 1.5 (black_box_wrapper.py:104-105)
    tau = np.clip(params[:, 0], 0.8, 1.5)
    delay = np.clip(params[:, 1], 0.05, 0.15)
    ip = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    iv = np.zeros((B, D), np.float32)
    pos, vel, y, v, h, slope = _prodmp_case(pc, bc, tc, 0.008, 2.8, params, ip, iv, 0.0, tau=tau, delay=delay,
                                            factory=factory)
    _check(pos, vel, y, v, h, bc.alpha)
    # against the ODE at the EXACT sample times: + half a table step of slope (index rounding)
    eng = (factory or make_engine)(pc, bc, tc, 0.008, 2.8)
    base = eng.times().astype(np.float64)
    cen, bw = R.rbf_constants(eng)
    tabs = eng.prodmp_tables()
    sc = tabs["scale"].astype(np.float32).astype(np.float64)
    for b in (2, 3):
        w = params[b, 2:].reshape(D, 3).astype(np.float64) * sc[:3] * np.float32(0.7)
        g = ip[b].astype(np.float64)                                  # disabled goal + relative goal: goal = init_pos
        s = np.maximum((base - delay[b]) / tau[b], 0.0)
        ye, _ = R.solve(25.0, 3.0, cen, bw, w, g, 0.0, ip[b].astype(np.float64), np.zeros(D), s)
        bound = slope[b] * h / 2 + (40 * h * h + 1e-5) * np.abs(ye).max()
        assert np.abs(pos[b] - ye.T).max() <= bound


def test_cfg3_dmp_euler_on_the_gpu_converges_to_the_ode_first_order():
    check_cfg3_dmp()


def check_cfg3_dmp(factory=None):
    """explicit Euler: the trajectory approaches the ODE solution linearly in dt"""
    pc, bc, tc, _, dur = CFG3
    B, D = 4, 7
    rng = np.random.default_rng(6)
    params = (rng.standard_normal((B, 42)) * 20.0).astype(np.float32)
    params[:, 5::6] = rng.standard_normal((B, 7))          # goals O(1)
    ip = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    iv = np.zeros((B, D), np.float32)
    errs = []
    for dt in (0.02, 0.01, 0.005):
        eng = (factory or make_engine)(pc, bc, tc, dt, dur)
        pos, vel = eng.trajectory(params, ip, iv, 0.0)
        pos = pos.cpu().numpy().astype(np.float64)
        cen, bw = R.rbf_constants(eng)
        t = eng.times().astype(np.float64)
        s = t / 4.0
        local = params.reshape(B, D, 6).astype(np.float64)
        e = 0.0
        for b in range(B):
            # the first sample carries the initial condition (MPK_DMP_FIRST_IS_INIT): the boundary is at s[0]
            y, _ = R.solve(25.0, 2.0, cen, bw, local[b, :, :5], local[b, :, 5], s[0], ip[b].astype(np.float64),
                           np.zeros(D), s)
            e = max(e, np.abs(pos[b] - y.T).max())
        errs.append(e)
    assert errs[1] < 0.6 * errs[0] and errs[2] < 0.6 * errs[1], errs
    assert errs[2] < 0.05 * np.abs(pos).max(), errs
