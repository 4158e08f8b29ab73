"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in ``fancy_gym_amd/`` may import this file.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as the
checker / the timed CPU baseline -- never as the product path.

PARITY UNPINNED
---------------
The arithmetic of the hot path lives in the third-party package ``mp_pytorch`` (pinned ``<=0.1.3`` by the reference:
``pyproject.toml:30``, ``setup.py:63``), which is NOT vendored under ``/root/reference`` and is not installable in the
build container (no network).  This file is therefore a *restatement* of that package's published algorithm
(Ijspeert et al. DMP; Paraschos et al. ProMP; Li et al. 2023 ProDMP) as recalled from the public
``ALRhub/MP_PyTorch`` v0.1.x sources, anchored on the reference's own call sites, behavioural tests and docs:

* call sites  : ``fancy_gym/black_box/black_box_wrapper.py:57,62-65,96-120,124-125,226``
* factories   : ``fancy_gym/black_box/factory/{phase,basis,trajectory}_generator_factory.py``
* controllers : ``fancy_gym/black_box/controller/*.py`` (these ARE in the reference; restated 1:1)
* rollout     : ``fancy_gym/black_box/black_box_wrapper.py:150-217``
* pins        : ``test/test_black_box.py:168-193,219-368`` (param count/order, plateaus),
                ``test/test_replanning_sequencing.py:64-364`` (lengths, planning counts),
                ``test/test_controller.py:30-54`` (PD formula exactness, shape errors)

HOW TO PIN IT: ``python tools/pin_against_mp_pytorch.py`` on a machine where ``mp_pytorch`` is importable drives the package
through the reference's own factory + ``BlackBoxWrapper.get_trajectory`` call sequence for the five BASELINE
configurations, TableTennis-ProDMP and one probe per "(?)" switch, writes ``tests/golden/ref_*.npz`` and reports which
switch settings reproduce it; ``tests/test_oracle_pins.py::test_oracle_against_mp_pytorch_reference_outputs`` then runs
from those files everywhere (it SKIPS, saying so, until they exist).

What pins it until then (see tests/test_oracle_*.py and tests/golden/make_golden.py):
  1. the reference's behavioural tests, re-stated against this oracle (plateaus with ``==``, action-dim formula,
     PD formula with ``array_equal``);
  2. an independent SciPy ODE solve of the DMP/ProDMP dynamics (closed form vs ``solve_ivp``);
  3. a second, independent formulation written with real ``torch`` CPU fp32 ops in the op order recalled from
     mp_pytorch (tests/golden/make_golden.py), whose outputs are committed as fixtures under tests/golden/.

Every function takes ``dtype``: ``np.float32`` mirrors the reference's default precision and op order (one rounding per
numpy op, no FMA); ``np.float64`` is the "true value" used to judge both.
All trajectory functions are batched over a leading ``B`` (the reference itself always runs B = 1:
``black_box_wrapper.py:96-120``).
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence, Tuple, Union

import numpy as np

Array = np.ndarray


# ----------------------------------------------------------------------------------------------------------------------
# configuration records (mirror the kwarg groups of fancy_gym/envs/registry.py:62-129)
# ----------------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class PhaseCfg:
    """phase_generator_kwargs (factory/phase_generator_factory.py:9-23)."""
    phase_generator_type: str = "linear"   # 'linear' | 'exp'
    tau: float = 3.0
    delay: float = 0.0
    alpha_phase: float = 3.0
    learn_tau: bool = False
    learn_delay: bool = False
    tau_bound: Tuple[float, float] = (1e-5, np.inf)
    delay_bound: Tuple[float, float] = (0.0, np.inf)


@dataclasses.dataclass
class BasisCfg:
    """basis_generator_kwargs (factory/basis_generator_factory.py:8-23)."""
    basis_generator_type: str = "rbf"      # 'rbf' | 'zero_rbf' | 'prodmp'
    num_basis: int = 10
    basis_bandwidth_factor: float = 3.0
    num_basis_outside: int = 0
    num_basis_zero_start: int = 2          # zero_rbf only (mp_pytorch default 2; fancy_gym sets 1)
    num_basis_zero_goal: int = 0           # zero_rbf only
    alpha: float = 25.0                    # prodmp only
    dt: float = 0.01                       # prodmp pre-compute grid step (the reference never overrides it)
    pre_compute_length_factor: int = 6     # prodmp only
    # SURVEY A.4 "(?)": ONE radial basis function in total has no neighbouring centre to take a gap from.
    # 'unit_gap' (default): bandwidth = factor / 1**2;  'refuse': ValueError (mpk.h MPK_SINGLE_RBF_*)
    single_rbf_mode: str = "unit_gap"


@dataclasses.dataclass
class TrajCfg:
    """trajectory_generator_kwargs (factory/trajectory_generator_factory.py:7-21)."""
    trajectory_generator_type: str = "promp"   # 'promp' | 'dmp' | 'prodmp'
    action_dim: int = 1
    weights_scale: float = 1.0
    goal_scale: float = 1.0
    alpha: float = 25.0                    # dmp spring constant (beta = alpha / 4)
    auto_scale_basis: bool = False
    relative_goal: bool = False
    disable_goal: bool = False
    disable_weights: bool = False
    # SURVEY Appendix A "(?)" items as explicit switches (same names and meaning as include/mpk.h); first = default
    relative_goal_mode: str = "before_scale"  # 'before_scale': goal = s_g*(g + init_pos) | 'after_scale': s_g*g + init_pos
    #   ^ UNPINNED either way.  Default flipped in round 4 (was 'after_scale'): three independent readers of upstream
    #     prodmp.py recall init_pos joining the RAW goal parameter; no reference config distinguishes the two at 1e-5;
    #     tools/pin_against_mp_pytorch.py decides it where mp_pytorch is installed (include/mpk.h MPK_RELGOAL_*)
    goal_offset_mode: str = "ignore"          # 'ignore' (swallowed by **kwargs) | 'add': goal += goal_offset
    goal_offset: float = 0.0
    dmp_first_sample: str = "init"            # 'init': pos[0] = init_pos | 'step': pos[0] = one Euler step from init_time


# ----------------------------------------------------------------------------------------------------------------------
# times (black_box_wrapper.py:57,115)
# ----------------------------------------------------------------------------------------------------------------------
def linspace32(start: float, end: float, steps: int) -> Array:
    """
    fp32 ``linspace`` following torch's documented symmetric scalar algorithm
    (step = (end-start)/(steps-1); i < steps//2 -> start + step*i, else end - step*(steps-1-i)), one fp32 rounding
    per op, no FMA.  torch's *vectorised* CPU kernel deviates from this by <= 1 ulp in a machine-dependent way
    (AVX2 vs AVX-512 chunking); tests/golden/make_golden.py records that the rounded ProDMP table indices are
    identical for every BASELINE configuration.
    """
    f = np.float32
    if steps == 1:
        return np.array([start], dtype=f)
    start, end = f(start), f(end)
    step = f((end - start) / f(steps - 1))
    i = np.arange(steps)
    lo = (start + step * i.astype(f)).astype(f)
    hi = (end - step * (steps - 1 - i).astype(f)).astype(f)
    return np.where(i < steps // 2, lo, hi).astype(f)


def _linspace(start, end, steps: int, dtype) -> Array:
    """fp32 -> the torch-style recipe above; fp64 -> numpy's."""
    if dtype == np.float32:
        return linspace32(start, end, steps)
    return np.linspace(start, end, steps, dtype=np.float64)


def num_steps(duration: float, dt: float) -> int:
    """T = round(duration / dt)  (python round = half-to-even, as the reference's ``round`` call)."""
    return int(round(duration / dt))


def make_times(duration: Optional[float], dt: float, init_time: Union[float, Array] = 0.0,
               tau_for_none: Optional[float] = None, dtype=np.float32) -> Array:
    """
    ``set_duration(duration, dt)``: times = linspace(0, duration, round(duration/dt)+1)[1:] + init_time.
    ``duration=None`` (sub-trajectory mode, black_box_wrapper.py:98-102) uses round(tau/dt)*dt
    (length pinned by test/test_replanning_sequencing.py:104-105).
    Returns [T] for scalar init_time, [B, T] for init_time of shape [B].
    """
    if duration is None:
        assert tau_for_none is not None
        duration = round(float(tau_for_none) / dt) * dt
    n = num_steps(duration, dt)
    # The reference's time grid is an fp32 tensor (init_time becomes an fp32 tensor in set_initial_conditions and the
    # sum is an fp32 op); the grid feeds the ProDMP table indices -- the integer part of the path -- so it is the fp32
    # values in every `dtype` (a float64 sum rounds differently and can move an index across a .5 tie).
    base = linspace32(0.0, duration, n + 1)[1:]
    it = np.asarray(init_time, dtype=np.float32)
    if it.ndim == 0:
        return (base + it).astype(np.float32).astype(dtype)
    return (base[None, :] + it[:, None]).astype(np.float32).astype(dtype)


# ----------------------------------------------------------------------------------------------------------------------
# phase generators
# ----------------------------------------------------------------------------------------------------------------------
def _q(x, dtype):
    """config scalar as the reference holds it: an fp32 tensor / python float mixed into fp32 tensor ops."""
    return dtype(np.float32(x))


def _col(x, dtype):
    """scalar or [B] -> broadcastable against [..., T]."""
    x = np.asarray(x, dtype=dtype)
    return x if x.ndim == 0 else x[:, None]


def linear_unbound(times: Array, tau, delay, dtype=np.float32) -> Array:
    return ((times.astype(dtype) - _col(delay, dtype)) / _col(tau, dtype)).astype(dtype)


def linear_left_bound(times: Array, tau, delay, dtype=np.float32) -> Array:
    return np.maximum(linear_unbound(times, tau, delay, dtype), dtype(0)).astype(dtype)


def phase(pc: PhaseCfg, times: Array, tau=None, delay=None, dtype=np.float32) -> Array:
    """bounded phase: linear -> clip(s,0,1); exp -> exp(-alpha_phase * max(s,0))."""
    tau = _q(pc.tau, dtype) if tau is None else tau
    delay = _q(pc.delay, dtype) if delay is None else delay
    if pc.phase_generator_type == "linear":
        return np.clip(linear_unbound(times, tau, delay, dtype), dtype(0), dtype(1)).astype(dtype)
    if pc.phase_generator_type == "exp":
        return np.exp(-_q(pc.alpha_phase, dtype) * linear_left_bound(times, tau, delay, dtype)).astype(dtype)
    raise ValueError(pc.phase_generator_type)


def unbound_phase(pc: PhaseCfg, times: Array, tau=None, delay=None, dtype=np.float32) -> Array:
    tau = _q(pc.tau, dtype) if tau is None else tau
    delay = _q(pc.delay, dtype) if delay is None else delay
    s = linear_unbound(times, tau, delay, dtype)
    if pc.phase_generator_type == "linear":
        return s
    return np.exp(-_q(pc.alpha_phase, dtype) * s).astype(dtype)


# ----------------------------------------------------------------------------------------------------------------------
# normalised RBF bases
# ----------------------------------------------------------------------------------------------------------------------
def rbf_total_basis(bc: BasisCfg) -> int:
    if bc.basis_generator_type == "zero_rbf":
        return bc.num_basis + bc.num_basis_zero_start + bc.num_basis_zero_goal
    return bc.num_basis


def rbf_centers_bandwidth(pc: PhaseCfg, bc: BasisCfg, dtype=np.float32) -> Tuple[Array, Array]:
    """
    Centres equally spaced in TIME over [delay, delay+tau] (+- outside), mapped through the UNBOUND phase with the
    construction-time tau/delay; bandwidth_k = factor / gap_k^2 with the last gap repeated.
    A single basis (no gap exists) uses gap = 1 phase unit -- unspecified upstream (?), only its action-space size is
    pinned (test/test_black_box.py:168-193).
    """
    n = rbf_total_basis(bc)
    if n == 1 and bc.single_rbf_mode == "refuse" and bc.basis_generator_type != "prodmp":
        raise ValueError("a single radial basis function has no neighbouring centre to take its bandwidth from")
    out = 0 if bc.basis_generator_type == "zero_rbf" else bc.num_basis_outside
    tau, delay = _q(pc.tau, dtype), _q(pc.delay, dtype)
    dist = tau / dtype(n - 2 * out - 1) if n > 1 else tau
    lo = dtype(-out) * dist + delay
    hi = tau + dtype(out) * dist + delay
    centers_t = _linspace(lo, hi, n, dtype)
    centers_p = unbound_phase(pc, centers_t, dtype=dtype)
    if n > 1:
        gaps = np.concatenate([centers_p[1:] - centers_p[:-1], centers_p[-1:] - centers_p[-2:-1]]).astype(dtype)
    else:
        gaps = np.ones(1, dtype=dtype)
    bw = (_q(bc.basis_bandwidth_factor, dtype) / (gaps * gaps)).astype(dtype)
    return centers_p.astype(dtype), bw


def rbf_basis(pc: PhaseCfg, bc: BasisCfg, times: Array, tau=None, delay=None, dtype=np.float32) -> Array:
    """[..., T] -> [..., T, n_total] normalised RBFs (all columns, incl. the zero-padding ones)."""
    c, bw = rbf_centers_bandwidth(pc, bc, dtype)
    x = phase(pc, times, tau, delay, dtype)[..., None]
    tmp = ((x - c) ** 2 * bw).astype(dtype)
    b = np.exp(-tmp / dtype(2)).astype(dtype)
    if c.shape[0] > 1:
        b = (b / b.sum(axis=-1, keepdims=True, dtype=dtype)).astype(dtype)
    return b


# ----------------------------------------------------------------------------------------------------------------------
# ProDMP basis tables (construction-time pre-compute)
# ----------------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class ProDMPTables:
    scaled_dt: float          # dt_basis / tau0, in `dtype`
    y1: Array                 # [N]
    y2: Array
    dy1: Array
    dy2: Array
    pos_basis: Array          # [N, nb+1]  (weights..., goal)
    vel_basis: Array          # [N, nb+1]
    scale_factors: Array      # [nb+1] = 1 / max_n pos_basis[n, k]  (auto_scale_basis)


def prodmp_tables(pc: PhaseCfg, bc: BasisCfg, dtype=np.float32) -> ProDMPTables:
    assert pc.phase_generator_type == "exp", "prodmp basis needs the exp phase (basis_generator_factory.py:16)"
    f = dtype
    alpha = _q(bc.alpha, f)
    tau0 = _q(pc.tau, f)
    scaled_dt = f(np.float32(bc.dt) / np.float32(pc.tau))   # an fp32 tensor op in the reference, in every dtype
    n_pc = int(bc.pre_compute_length_factor * int(np.round(1.0 / float(scaled_dt)))) + 1
    s = _linspace(0.0, bc.pre_compute_length_factor, n_pc, f)
    half = f(0.5) * alpha
    y1 = np.exp(-half * s).astype(f)
    y2 = (s * y1).astype(f)
    dy1 = (-half * y1).astype(f)
    dy2 = (-half * y2 + y1).astype(f)
    e = np.exp(half * s).astype(f)
    q1 = ((half * s - f(1)) * e + f(1)).astype(f)
    q2 = (half * (e - f(1))).astype(f)
    # RBFs and canonical phase on the grid mapped back to time with the construction-time tau/delay
    pc_times = (s * tau0 + _q(pc.delay, f)).astype(f)
    rb = BasisCfg(**{**dataclasses.asdict(bc), "basis_generator_type": "rbf"})
    phi = rbf_basis(pc, rb, pc_times, dtype=f)                        # [N, nb]
    x = phase(pc, pc_times, dtype=f)                                  # [N]
    dp1 = ((s * e * x)[:, None] * phi).astype(f)
    dp2 = ((e * x)[:, None] * phi).astype(f)
    # cumulative trapezoid: p[i] = trapz(dp[:i+1], s[:i+1])
    ds = np.diff(s).astype(f)[:, None]
    inc1 = (ds * (dp1[1:] + dp1[:-1]) / f(2)).astype(f)
    inc2 = (ds * (dp2[1:] + dp2[:-1]) / f(2)).astype(f)
    p1 = np.concatenate([np.zeros((1, phi.shape[1]), f), np.cumsum(inc1, axis=0, dtype=f)]).astype(f)
    p2 = np.concatenate([np.zeros((1, phi.shape[1]), f), np.cumsum(inc2, axis=0, dtype=f)]).astype(f)
    pos_w = (p2 * y2[:, None] - p1 * y1[:, None]).astype(f)
    pos_g = (q2 * y2 - q1 * y1).astype(f)
    vel_w = (p2 * dy2[:, None] - p1 * dy1[:, None]).astype(f)
    vel_g = (q2 * dy2 - q1 * dy1).astype(f)
    pos_basis = np.concatenate([pos_w, pos_g[:, None]], axis=1).astype(f)
    vel_basis = np.concatenate([vel_w, vel_g[:, None]], axis=1).astype(f)
    scale = (f(1) / pos_basis.max(axis=0)).astype(f)
    return ProDMPTables(float(scaled_dt), y1, y2, dy1, dy2, pos_basis, vel_basis, scale)


def prodmp_indices(times: Array, tau, delay, scaled_dt: float, length_factor: int = 6, dtype=np.float32) -> Array:
    """
    times_to_indices: idx = round_half_even( max((t-delay)/tau, 0) / scaled_dt ).  INTEGER output -- the bit-exact
    part of the path.  Raises if scaled time exceeds the pre-computed range.
    """
    s = linear_left_bound(times, tau, delay, dtype)
    if s.size and float(s.max()) > length_factor:
        raise RuntimeError("Time is beyond the pre-computation range. Set larger pre-computation factor")
    return np.rint((s / dtype(scaled_dt)).astype(dtype)).astype(np.int64)


# ----------------------------------------------------------------------------------------------------------------------
# parameter layout
# ----------------------------------------------------------------------------------------------------------------------
def num_basis_learnable(bc: BasisCfg) -> int:
    return bc.num_basis


def num_local_params(tc: TrajCfg, bc: BasisCfg) -> int:
    d, nb = tc.action_dim, bc.num_basis
    t = tc.trajectory_generator_type
    if t == "promp":
        return d * nb
    if t == "dmp":
        return d * (nb + 1)
    if t == "prodmp":
        k = (0 if tc.disable_weights else nb) + (0 if tc.disable_goal else 1)
        return d * k
    raise ValueError(t)


def num_params(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg) -> int:
    return int(pc.learn_tau) + int(pc.learn_delay) + num_local_params(tc, bc)


def params_bounds(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg) -> Array:
    """[2, P] float32: tau in tau_bound, delay in delay_bound, local params +-inf (black_box_wrapper.py:122-127)."""
    lo, hi = [], []
    if pc.learn_tau:
        lo.append(pc.tau_bound[0]); hi.append(pc.tau_bound[1])
    if pc.learn_delay:
        lo.append(pc.delay_bound[0]); hi.append(pc.delay_bound[1])
    n = num_local_params(tc, bc)
    lo += [-np.inf] * n
    hi += [np.inf] * n
    return np.array([lo, hi], dtype=np.float32)


def split_params(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg, params: Array, dtype=np.float32):
    """params [B, P] -> (tau [B] or scalar, delay [B] or scalar, local [B, D, K])."""
    params = np.asarray(params, dtype=dtype)
    assert params.ndim == 2 and params.shape[1] == num_params(pc, bc, tc), (params.shape, num_params(pc, bc, tc))
    i = 0
    tau, delay = _q(pc.tau, dtype), _q(pc.delay, dtype)
    if pc.learn_tau:
        tau = params[:, i]; i += 1
        assert tau.min() > 0
    if pc.learn_delay:
        delay = params[:, i]; i += 1
        assert delay.min() >= 0
    local = params[:, i:].reshape(params.shape[0], tc.action_dim, -1)
    return tau, delay, local


# ----------------------------------------------------------------------------------------------------------------------
# ProMP
# ----------------------------------------------------------------------------------------------------------------------
def promp_trajectory(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg, params: Array, times: Array, init_pos: Array,
                     dtype=np.float32) -> Tuple[Array, Array]:
    """
    pos = (Phi * weights_scale) . w^T  (+ init_pos when the basis is zero-padded); vel = forward difference of pos over
    times with the last row repeated.  times [T] or [B, T]; returns pos, vel [B, T, D].
    """
    f = dtype
    tau, delay, w = split_params(pc, bc, tc, params, f)            # w [B, D, nb]
    B = w.shape[0]
    phi = rbf_basis(pc, bc, times, tau, delay, f)                  # [T, n] or [B, T, n]
    if bc.basis_generator_type == "zero_rbf":
        zs = bc.num_basis_zero_start
        phi = phi[..., zs: zs + bc.num_basis]
    phi = (phi * _q(tc.weights_scale, f)).astype(f)
    if phi.ndim == 2:
        phi = np.broadcast_to(phi, (B,) + phi.shape)
    pos = np.einsum("btk,bdk->btd", phi, w).astype(f)
    if bc.basis_generator_type == "zero_rbf":
        pos = (pos + np.asarray(init_pos, f)[:, None, :]).astype(f)
    tt = np.broadcast_to(times.astype(f), (B, times.shape[-1])) if times.ndim == 1 else times.astype(f)
    vel = np.zeros_like(pos)
    vel[:, :-1] = (np.diff(pos, axis=1) / np.diff(tt, axis=1)[..., None]).astype(f)
    vel[:, -1] = vel[:, -2]
    return pos, vel


# ----------------------------------------------------------------------------------------------------------------------
# DMP
# ----------------------------------------------------------------------------------------------------------------------
def dmp_trajectory(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg, params: Array, times: Array, init_pos: Array,
                   init_vel: Array, dtype=np.float32, init_time=None) -> Tuple[Array, Array]:
    """
    forcing f = x * (Phi . w); explicit (semi-implicit) Euler in scaled time:
        a = alpha*(beta*(g - y_i) - z_i) + f_i ; z_{i+1} = z_i + ds_i*a ; y_{i+1} = y_i + ds_i*z_{i+1} ; vel = z/tau
    The first sample carries the initial condition (y_0 = init_pos, z_0 = tau*init_vel) -- ``dmp_first_sample='init'``,
    the default (SURVEY A.6 "(?)"); 'step': the first sample is one Euler step from (init_time, init_pos, init_vel).
    """
    f = dtype
    tau, delay, wg = split_params(pc, bc, tc, params, f)           # [B, D, nb+1]
    B, D = wg.shape[0], wg.shape[1]
    if tc.dmp_first_sample == "step":
        # the integration starts AT init_time (the time grid excludes it, A.2): prepend that sample, integrate, drop it
        assert init_time is not None
        it = np.broadcast_to(np.asarray(init_time, np.float32), (B,)).astype(np.float32).astype(f)
        tt = np.broadcast_to(times.astype(f), (B, times.shape[-1]))
        ext = np.concatenate([it[:, None], tt], axis=1).astype(f)
        sub = dataclasses.replace(tc, dmp_first_sample="init")
        pos, vel = dmp_trajectory(pc, bc, sub, params, ext, init_pos, init_vel, f)
        return pos[:, 1:], vel[:, 1:]
    w = (wg[..., :-1] * _q(tc.weights_scale, f)).astype(f)
    g = (wg[..., -1] * _q(tc.goal_scale, f)).astype(f)
    phi = rbf_basis(pc, bc, times, tau, delay, f)
    x = phase(pc, times, tau, delay, f)
    bx = (phi * x[..., None]).astype(f)
    if bx.ndim == 2:
        bx = np.broadcast_to(bx, (B,) + bx.shape)
    force = np.einsum("btk,bdk->btd", bx, w).astype(f)
    s = linear_left_bound(times, tau, delay, f)
    s = np.broadcast_to(s, (B, s.shape[-1])) if s.ndim == 1 else s
    ds = np.diff(s, axis=1).astype(f)                               # [B, T-1]
    T = s.shape[1]
    tau_b = np.broadcast_to(np.asarray(tau, f), (B,)).astype(f)
    alpha, beta = _q(tc.alpha, f), _q(tc.alpha / 4.0, f)
    pos = np.zeros((B, T, D), f)
    z = np.zeros((B, T, D), f)
    pos[:, 0] = np.asarray(init_pos, f)
    z[:, 0] = (np.asarray(init_vel, f) * tau_b[:, None]).astype(f)
    for i in range(T - 1):
        acc = (alpha * (beta * (g - pos[:, i]) - z[:, i]) + force[:, i]).astype(f)
        z[:, i + 1] = (z[:, i] + ds[:, i, None] * acc).astype(f)
        pos[:, i + 1] = (pos[:, i] + ds[:, i, None] * z[:, i + 1]).astype(f)
    vel = (z / tau_b[:, None, None]).astype(f)
    return pos, vel


# ----------------------------------------------------------------------------------------------------------------------
# ProDMP
# ----------------------------------------------------------------------------------------------------------------------
def prodmp_weights_goal_scale(tc: TrajCfg, bc: BasisCfg, tables: ProDMPTables, dtype=np.float32) -> Array:
    nb = bc.num_basis
    s = np.empty(nb + 1, dtype)
    if tc.auto_scale_basis:
        s[:] = tables.scale_factors.astype(dtype)
        s[:-1] = s[:-1] * _q(tc.weights_scale, dtype)
        s[-1] = s[-1] * _q(tc.goal_scale, dtype)
    else:
        s[:-1] = _q(tc.weights_scale, dtype)
        s[-1] = _q(tc.goal_scale, dtype)
    return s


def prodmp_full_params(tc: TrajCfg, bc: BasisCfg, local: Array, dtype=np.float32) -> Array:
    """[B, D, k] -> [B, D, nb+1] with zeros for the disabled goal / weights columns."""
    B, D, _ = local.shape
    nb = bc.num_basis
    full = np.zeros((B, D, nb + 1), dtype)
    c = 0
    if not tc.disable_weights:
        full[..., :nb] = local[..., :nb]; c = nb
    if not tc.disable_goal:
        full[..., nb] = local[..., c]
    return full


def prodmp_trajectory(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg, params: Array, times: Array,
                      init_time: Union[float, Array], init_pos: Array, init_vel: Array,
                      tables: Optional[ProDMPTables] = None, dtype=np.float32, return_indices: bool = False):
    """
    Closed form with boundary conditions at init_time (index b):
        det  = y1_b*dy2_b - y2_b*dy1_b
        xi1  = (dy2_b*y1 - dy1_b*y2)/det     xi2 = (y1_b*y2 - y2_b*y1)/det      (xi3, xi4: same with dy1, dy2)
        H    = Psi  - xi1 (x) Psi_b - xi2 (x) dPsi_b        Hv = dPsi - xi3 (x) Psi_b - xi4 (x) dPsi_b
        pos  = xi1*y_b + xi2*(tau*v_b) + H . wg            vel = (xi3*y_b + xi4*(tau*v_b) + Hv . wg) / tau
    wg = [w, g] * weights_goal_scale per DoF; relative_goal adds init_pos to the raw goal parameter -- or, with
    relative_goal_mode='after_scale', to the scaled goal (SURVEY A.5 "(?)"); goal_offset_mode='add' adds
    goal_offset to the goal (default: ignored, as **kwargs swallows it upstream).
    """
    f = dtype
    if tables is None:
        tables = prodmp_tables(pc, bc, f)
    tau, delay, local = split_params(pc, bc, tc, params, f)
    B, D = local.shape[0], local.shape[1]
    y_b = np.asarray(init_pos, f)
    full = prodmp_full_params(tc, bc, local, f)
    if tc.relative_goal and tc.relative_goal_mode == "before_scale":
        full[..., -1] = (full[..., -1] + y_b).astype(f)          # init_pos joins the RAW goal parameter
    wg = (full * prodmp_weights_goal_scale(tc, bc, tables, f)).astype(f)
    if tc.relative_goal and tc.relative_goal_mode == "after_scale":
        wg[..., -1] = (wg[..., -1] + y_b).astype(f)              # init_pos joins the scaled goal
    if tc.goal_offset_mode == "add":
        wg[..., -1] = (wg[..., -1] + _q(tc.goal_offset, f)).astype(f)
    tau_b = np.broadcast_to(np.asarray(tau, f), (B,)).astype(f)
    v_b = (np.asarray(init_vel, f) * tau_b[:, None]).astype(f)

    tt = np.broadcast_to(times.astype(f), (B, times.shape[-1])) if times.ndim == 1 else times.astype(f)
    # the table indices are the INTEGER part of the path: the reference computes them with fp32 tensor ops, so the
    # fp32 recipe defines them for every `dtype` (a float64 quotient can fall on the other side of a .5 tie)
    g = np.float32
    idx = prodmp_indices(tt.astype(g), np.asarray(tau, g), np.asarray(delay, g), tables.scaled_dt,
                         bc.pre_compute_length_factor, g)                                              # [B, T]
    it = np.broadcast_to(np.asarray(init_time, f), (B,)).astype(f)
    idx_b = prodmp_indices(it.astype(g)[:, None], np.asarray(tau, g), np.asarray(delay, g), tables.scaled_dt,
                           bc.pre_compute_length_factor, g)[:, 0]

    T1, T2, D1, D2 = (tables.y1.astype(f), tables.y2.astype(f), tables.dy1.astype(f), tables.dy2.astype(f))
    PB, VB = tables.pos_basis.astype(f), tables.vel_basis.astype(f)
    y1, y2, dy1, dy2 = T1[idx], T2[idx], D1[idx], D2[idx]                                               # [B, T]
    y1b, y2b, dy1b, dy2b = T1[idx_b], T2[idx_b], D1[idx_b], D2[idx_b]                                   # [B]
    det = (y1b * dy2b - y2b * dy1b).astype(f)
    a, b_, c, d = (dy2b / det)[:, None], (dy1b / det)[:, None], (y1b / det)[:, None], (y2b / det)[:, None]
    xi1 = (a * y1 - b_ * y2).astype(f)
    xi2 = (c * y2 - d * y1).astype(f)
    xi3 = (a * dy1 - b_ * dy2).astype(f)
    xi4 = (c * dy2 - d * dy1).astype(f)
    psi_b, dpsi_b = PB[idx_b], VB[idx_b]                                                                # [B, K]
    H = (PB[idx] - (xi1[..., None] * psi_b[:, None, :] + xi2[..., None] * dpsi_b[:, None, :])).astype(f)
    Hv = (VB[idx] - (xi3[..., None] * psi_b[:, None, :] + xi4[..., None] * dpsi_b[:, None, :])).astype(f)
    pos_lin = np.einsum("btk,bdk->btd", H, wg).astype(f)
    vel_lin = np.einsum("btk,bdk->btd", Hv, wg).astype(f)
    pos = (xi1[..., None] * y_b[:, None, :] + xi2[..., None] * v_b[:, None, :] + pos_lin).astype(f)
    vel = (xi3[..., None] * y_b[:, None, :] + xi4[..., None] * v_b[:, None, :] + vel_lin).astype(f)
    vel = (vel / tau_b[:, None, None]).astype(f)
    if return_indices:
        return pos, vel, idx, idx_b
    return pos, vel


# ----------------------------------------------------------------------------------------------------------------------
# one entry point = BlackBoxWrapper.get_trajectory (black_box_wrapper.py:96-120), batched
# ----------------------------------------------------------------------------------------------------------------------
def get_trajectory(pc: PhaseCfg, bc: BasisCfg, tc: TrajCfg, params: Array, duration: Optional[float], dt: float,
                   init_time: Union[float, Array], init_pos: Array, init_vel: Array, dtype=np.float32,
                   clip: bool = True, tables: Optional[ProDMPTables] = None) -> Tuple[Array, Array]:
    params = np.asarray(params, dtype)
    if params.ndim == 1:
        params = params[None]
    if clip:
        bnd = params_bounds(pc, bc, tc).astype(dtype)
        params = np.clip(params, bnd[0], bnd[1]).astype(dtype)
    B = params.shape[0]
    init_pos = np.broadcast_to(np.asarray(init_pos, dtype), (B, tc.action_dim))
    init_vel = np.broadcast_to(np.asarray(init_vel, dtype), (B, tc.action_dim))
    tau_none = None
    if duration is None:
        assert pc.learn_tau
        tau_none = float(np.max(params[:, 0]))
    times = make_times(duration, dt, init_time, tau_none, dtype)
    t = tc.trajectory_generator_type
    if t == "promp":
        return promp_trajectory(pc, bc, tc, params, times, init_pos, dtype)
    if t == "dmp":
        return dmp_trajectory(pc, bc, tc, params, times, init_pos, init_vel, dtype, init_time=init_time)
    if t == "prodmp":
        return prodmp_trajectory(pc, bc, tc, params, times, init_time, init_pos, init_vel, tables, dtype)
    raise ValueError(t)


# ----------------------------------------------------------------------------------------------------------------------
# tracking controllers (fancy_gym/black_box/controller/*.py -- these ARE reference code paths; restated 1:1)
# ----------------------------------------------------------------------------------------------------------------------
def pd_action(p_gains, d_gains, des_pos, des_vel, c_pos, c_vel):
    """pd_controller.py:21-29.  numpy promotion as in the reference (fp32 desired (+) fp64 state -> fp64)."""
    des_pos, des_vel, c_pos, c_vel = map(np.asarray, (des_pos, des_vel, c_pos, c_vel))
    if des_pos.shape != c_pos.shape:
        raise ValueError(f"Mismatch in dimension between desired position {des_pos.shape} and current position "
                         f"{c_pos.shape}")
    if des_vel.shape != c_vel.shape:
        raise ValueError(f"Mismatch in dimension between desired velocity {des_vel.shape} and current velocity "
                         f"{c_vel.shape}")
    return p_gains * (des_pos - c_pos) + d_gains * (des_vel - c_vel)


def pos_action(des_pos, des_vel, c_pos, c_vel):      # pos_controller.py:8-9
    return des_pos


def vel_action(des_pos, des_vel, c_pos, c_vel):      # vel_controller.py:8-9
    return des_vel


def metaworld_action(des_pos, des_vel, c_pos, c_vel):   # meta_world_controller.py:15-25
    des_pos, c_pos = np.asarray(des_pos), np.asarray(c_pos)
    xyz, cur = des_pos[..., :-1], c_pos[..., :-1]
    if xyz.shape != cur.shape:
        raise ValueError(f"Mismatch in dimension between desired position {xyz.shape} and current position {cur.shape}")
    return np.concatenate([xyz - cur, des_pos[..., -1:]], axis=-1)


CONTROLLERS = {"motor": "pd", "position": "pos", "velocity": "vel", "metaworld": "metaworld"}


# ----------------------------------------------------------------------------------------------------------------------
# rollout loop (black_box_wrapper.py:175-203) against the reference's only batchable plants
# ----------------------------------------------------------------------------------------------------------------------
def rollout(des_pos: Array, des_vel: Array, controller: str, p_gains, d_gains, act_low, act_high, plant: str,
            dt: float, pos0: Array, vel0: Array, n_steps: Optional[Array] = None):
    """
    Batched restatement of the per-step loop:  action_t = clip(controller(des_t, state_t), low, high); state <- plant.
    plant 'static'            : state never changes (test/test_black_box.py:50-56 ToyWrapper: pos == 1, vel == 0)
    plant 'double_integrator' : vel += dt*a ; pos += dt*vel   (envs/classic_control/base_reacher/
                                base_reacher_torque.py:25-26)
    All arithmetic in float64 (the reference's numpy promotion).  n_steps [B] optionally stops episode b after that
    many steps (actions beyond are 0) -- the replanning / termination break of black_box_wrapper.py:197.
    Returns actions [B, T, D] f64, final pos, final vel.
    """
    des_pos = np.asarray(des_pos); des_vel = np.asarray(des_vel)
    B, T, D = des_pos.shape
    q = np.array(np.broadcast_to(np.asarray(pos0, np.float64), (B, D)))
    qd = np.array(np.broadcast_to(np.asarray(vel0, np.float64), (B, D)))
    lo = np.asarray(act_low, np.float64); hi = np.asarray(act_high, np.float64)
    pg = np.asarray(p_gains, np.float64); dg = np.asarray(d_gains, np.float64)
    actions = np.zeros((B, T, D), np.float64)
    n = np.full(B, T) if n_steps is None else np.asarray(n_steps)
    for t in range(T):
        live = (t < n)[:, None]
        if controller == "motor":
            a = pg * (des_pos[:, t] - q) + dg * (des_vel[:, t] - qd)
        elif controller == "position":
            a = des_pos[:, t].astype(np.float64)
        elif controller == "velocity":
            a = des_vel[:, t].astype(np.float64)
        else:
            raise ValueError(controller)
        a = np.clip(a, lo, hi)
        actions[:, t] = np.where(live, a, 0.0)
        if plant == "double_integrator":
            qd_n = qd + dt * a
            q_n = q + dt * qd_n
            qd = np.where(live, qd_n, qd)
            q = np.where(live, q_n, q)
        elif plant != "static":
            raise ValueError(plant)
    return actions, q, qd


def reacher_rollout(des_pos: Array, des_vel: Array, controller: str, p_gains, d_gains, act_low, act_high, dt: float,
                    pos0: Array, vel0: Array, goal: Array, n_steps: Optional[Array] = None,
                    step0: Optional[Array] = None, steps_before_reward: int = 199):
    """
    `rollout` on the torque double integrator plus the reward of the reference's SimpleReacher:
      plant   BaseReacherTorqueEnv.step, envs/classic_control/base_reacher/base_reacher_torque.py:20-37
              (vel += dt*a; pos += dt*vel; then reward; then the step counter advances)
      joints  BaseReacherEnv._update_joints, base_reacher/base_reacher.py:97-104: angles = cumsum(q),
              end effector = cumsum over the links of link_length * (cos, sin), link lengths 1 (:19)
      reward  SimpleReacherEnv._get_reward, simple_reacher/simple_reacher.py:56-72:
              -(||ee - goal|| if steps >= steps_before_reward else 0) - sum(action ** 2), steps_before_reward = 199 (:30)
    `action` is what BlackBoxWrapper passes to env.step: the clipped controller output (black_box_wrapper.py:176-181).
    goal [B, 2]; step0 [B] = env step counter when this call starts.  float64.
    Returns actions [B, T, D], rewards [B, T], final pos, final vel.
    """
    des_pos = np.asarray(des_pos); des_vel = np.asarray(des_vel)
    B, T, D = des_pos.shape
    q = np.array(np.broadcast_to(np.asarray(pos0, np.float64), (B, D)))
    qd = np.array(np.broadcast_to(np.asarray(vel0, np.float64), (B, D)))
    goal = np.array(np.broadcast_to(np.asarray(goal, np.float64), (B, 2)))
    lo = np.asarray(act_low, np.float64); hi = np.asarray(act_high, np.float64)
    pg = np.asarray(p_gains, np.float64); dg = np.asarray(d_gains, np.float64)
    actions = np.zeros((B, T, D), np.float64)
    rewards = np.zeros((B, T), np.float64)
    n = np.full(B, T) if n_steps is None else np.asarray(n_steps)
    s0 = np.zeros(B, np.int64) if step0 is None else np.asarray(step0, np.int64)
    for t in range(T):
        live = t < n
        if controller == "motor":
            a = pg * (des_pos[:, t] - q) + dg * (des_vel[:, t] - qd)
        elif controller == "position":
            a = des_pos[:, t].astype(np.float64)
        elif controller == "velocity":
            a = des_vel[:, t].astype(np.float64)
        else:
            raise ValueError(controller)
        a = np.clip(a, lo, hi)
        qd_n = qd + dt * a
        q_n = q + dt * qd_n
        angles = np.cumsum(q_n, axis=1)
        ee = np.stack([np.cumsum(np.cos(angles), axis=1)[:, -1], np.cumsum(np.sin(angles), axis=1)[:, -1]], axis=1)
        diff = ee - goal
        dist = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
        ctrl = np.zeros(B)
        for d in range(D):                      # (action ** 2).sum() of a short vector: left to right
            ctrl = a[:, d] * a[:, d] if d == 0 else ctrl + a[:, d] * a[:, d]
        r = np.where(s0 + t >= steps_before_reward, 0.0 - dist, 0.0) - ctrl
        actions[:, t] = np.where(live[:, None], a, 0.0)
        rewards[:, t] = np.where(live, r, 0.0)
        qd = np.where(live[:, None], qd_n, qd)
        q = np.where(live[:, None], q_n, q)
    return actions, rewards, q, qd


def traj_invalid_penalty(action: Array, pos_traj: Array, pos_low: Array, pos_high: Array, tau_bound=None,
                         delay_bound=None) -> Array:
    """
    Reward of a plan that fails the validity check and is not executed (BlackBoxWrapper.step -> invalid_traj_callback,
    black_box_wrapper.py:169-172; TableTennisEnv._get_traj_invalid_penalty, envs/mujoco/table_tennis/
    table_tennis_env.py:282-289), batched: action [B, P] (raw, unclipped; action[:, 0] = tau, action[:, 1] = delay),
    pos_traj [B, T, D].  float64.  tau_bound / delay_bound None drops those terms.
    """
    action = np.asarray(action, np.float64); pos = np.asarray(pos_traj, np.float64)
    pen = np.zeros(pos.shape[0])
    if tau_bound is not None and delay_bound is not None:
        pen = pen + 3 * (np.maximum(0, action[:, 0] - tau_bound[1]) + np.maximum(0, tau_bound[0] - action[:, 0]))
        pen = pen + 3 * (np.maximum(0, action[:, 1] - delay_bound[1]) + np.maximum(0, delay_bound[0] - action[:, 1]))
    pen = pen + np.mean(np.maximum(pos - np.asarray(pos_high, np.float64), 0), axis=(1, 2))
    pen = pen + np.mean(np.maximum(np.asarray(pos_low, np.float64) - pos, 0), axis=(1, 2))
    return -pen


# ----------------------------------------------------------------------------------------------------------------------
# replanning bookkeeping (black_box_wrapper.py:107-108,174,197-206) -- INTEGER state, bit-exact part
# ----------------------------------------------------------------------------------------------------------------------
def replanning_segments(horizon: int, every: int, max_planning_times: int):
    """
    For the schedule ``t % every == 0`` (e.g. box_pushing/mp_wrapper.py:89) return per plan k the tuple
    (current_traj_steps at plan start, trajectory_length executed).  The k-th call breaks at local step t when
    (t + 1 + current_traj_steps) % every == 0 and plan_steps < max_planning_times, or when the env truncates at
    `horizon` global steps.
    """
    out, cur, plan = [], 0, 0
    while cur < horizon:
        plan += 1
        length = 0
        for t in range(horizon):            # generated trajectory always has `horizon` samples
            length = t + 1
            g = t + 1 + cur
            if g >= horizon:
                break
            if g % every == 0 and plan < max_planning_times:
                break
        out.append((cur, length))
        cur += length
    return out
