#!/usr/bin/env python3
"""
Generates tests/golden/*.npz -- run in the BUILD container only (it needs nothing from /root/reference: the
reference's arithmetic lives in the un-vendored mp_pytorch package, which cannot be imported here; see
oracle/mp_oracle.py "PARITY UNPINNED").

What is stored per BASELINE configuration (small batches, a few KB each):
  inputs      params, init_pos, init_vel, init_time (seeded)
  oracle32    pos / vel from oracle/mp_oracle.py in float32 (reference precision and op order)
  oracle64    the same in float64 (the "true value" of the restated maths)
  torch32     pos / vel from a SECOND, independent formulation written below with real torch CPU float32 tensor ops
              in the op order recalled from mp_pytorch (class-per-concept, pre-computed ProDMP tables by a
              torch.trapz loop, einsum contractions, torch.linspace time grid)
  idx         ProDMP table indices (int32) from both formulations -- must be identical
  actions     PD controller actions of the reference formula on (pos, vel) for the two GPU-resident plants
The metadata says so explicitly: these vectors come from restatements, NOT from mp_pytorch.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mp_oracle as O  # noqa: E402

torch.set_num_threads(1)
F = torch.float32


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


GENERATOR_SHA256 = _sha256(os.path.abspath(__file__))
ORACLE_SHA256 = _sha256(os.path.join(ROOT, "oracle", "mp_oracle.py"))


# ======================================================================================================================
# second formulation: torch CPU fp32, mp_pytorch-style objects
# ======================================================================================================================
class TPhase:
    def __init__(self, kind, tau, delay=0.0, alpha_phase=3.0):
        self.kind = kind
        self.tau = torch.tensor(tau).float()
        self.delay = torch.tensor(delay).float()
        self.alpha_phase = torch.tensor(alpha_phase).float()

    def left_bound_linear_phase(self, times):
        return torch.clip((times - self.delay[..., None]) / self.tau[..., None], min=0)

    def unbound_linear(self, times):
        return (times - self.delay[..., None]) / self.tau[..., None]

    def phase(self, times):
        if self.kind == "linear":
            return torch.clip((times - self.delay[..., None]) / self.tau[..., None], 0, 1)
        return torch.exp(-self.alpha_phase * self.left_bound_linear_phase(times))

    def unbound_phase(self, times):
        if self.kind == "linear":
            return self.unbound_linear(times)
        return torch.exp(-self.alpha_phase * self.unbound_linear(times))


class TRBF:
    def __init__(self, phase: TPhase, num_basis, bandwidth=3.0, outside=0):
        self.pg, self.n = phase, num_basis
        dist = phase.tau / (num_basis - 2 * outside - 1) if num_basis > 1 else phase.tau
        centers_t = torch.linspace(float(-outside * dist + phase.delay), float(phase.tau + outside * dist + phase.delay),
                                   num_basis)
        self.centers_p = phase.unbound_phase(centers_t)
        tmp = torch.cat((self.centers_p[1:] - self.centers_p[:-1], self.centers_p[-1:] - self.centers_p[-2:-1]), dim=-1)
        self.bandWidth = bandwidth / (tmp ** 2)

    def basis(self, times):
        phase = self.pg.phase(times)[..., None].expand([*times.shape, self.n])
        centers = self.centers_p[None, :].expand([times.shape[-1], -1])
        tmp = torch.einsum('...ij,...j->...ij', (phase - centers) ** 2, self.bandWidth)
        basis = torch.exp(-tmp / 2)
        if self.n > 1:
            basis = basis / torch.sum(basis, dim=-1, keepdim=True)
        return basis


class TProDMPBasis(TRBF):
    def __init__(self, phase, num_basis, bandwidth, alpha, dt=0.01, factor=6):
        super().__init__(phase, num_basis, bandwidth)
        self.alpha = alpha
        self.scaled_dt = torch.tensor(dt).float() / phase.tau
        self.factor = factor
        n_pc = factor * int(torch.round(1 / self.scaled_dt).long().item()) + 1
        s = torch.linspace(0, factor, n_pc)
        self.y1 = torch.exp(-0.5 * alpha * s)
        self.y2 = s * self.y1
        self.dy1 = -0.5 * alpha * self.y1
        self.dy2 = -0.5 * alpha * self.y2 + self.y1
        q1 = (0.5 * alpha * s - 1) * torch.exp(0.5 * alpha * s) + 1
        q2 = 0.5 * alpha * (torch.exp(0.5 * alpha * s) - 1)
        pc_times = s * phase.tau + phase.delay
        basis_single = super().basis(pc_times)
        x = phase.phase(pc_times)
        dp1 = torch.einsum('...i,...i,...ik->...ik', s * torch.exp(alpha * s / 2), x, basis_single)
        dp2 = torch.einsum('...i,...i,...ik->...ik', torch.exp(alpha * s / 2), x, basis_single)
        p1, p2 = torch.zeros_like(dp1), torch.zeros_like(dp2)
        for i in range(s.shape[0]):
            p1[i] = torch.trapz(dp1[:i + 1], s[:i + 1], dim=0)
            p2[i] = torch.trapz(dp2[:i + 1], s[:i + 1], dim=0)
        pos_w = p2 * self.y2[:, None] - p1 * self.y1[:, None]
        pos_g = q2 * self.y2 - q1 * self.y1
        vel_w = p2 * self.dy2[:, None] - p1 * self.dy1[:, None]
        vel_g = q2 * self.dy2 - q1 * self.dy1
        self.pc_pos = torch.cat([pos_w, pos_g[:, None]], dim=-1)
        self.pc_vel = torch.cat([vel_w, vel_g[:, None]], dim=-1)
        self.scale_factors = 1. / self.pc_pos.max(dim=0).values

    def indices(self, times, tau, delay):
        scaled = torch.clip((times - delay[..., None]) / tau[..., None], min=0)
        return torch.round(scaled / self.scaled_dt).long()


def t_times(duration, dt, init_time):
    times = torch.linspace(0, duration, round(duration / dt) + 1, dtype=F)
    return times[1:] + torch.as_tensor(init_time, dtype=F)[..., None]


def t_split(pc: O.PhaseCfg, params, D):
    params = torch.as_tensor(params, dtype=F)
    i = 0
    tau, delay = torch.tensor(pc.tau).float().expand(params.shape[0]), torch.tensor(pc.delay).float().expand(params.shape[0])
    if pc.learn_tau:
        tau = params[:, i]; i += 1
    if pc.learn_delay:
        delay = params[:, i]; i += 1
    return tau, delay, params[:, i:].reshape(params.shape[0], D, -1)


def t_promp(pc, bc, tc, params, duration, dt, init_time, init_pos):
    tau, delay, w = t_split(pc, params, tc.action_dim)
    B = w.shape[0]
    times = t_times(duration, dt, np.broadcast_to(init_time, (B,)))
    zs = bc.num_basis_zero_start if bc.basis_generator_type == "zero_rbf" else 0
    zg = bc.num_basis_zero_goal if bc.basis_generator_type == "zero_rbf" else 0
    rbf = TRBF(TPhase(pc.phase_generator_type, pc.tau, pc.delay, pc.alpha_phase), bc.num_basis + zs + zg,
               bc.basis_bandwidth_factor, 0 if zs + zg else bc.num_basis_outside)
    pos = torch.zeros(B, times.shape[-1], tc.action_dim)
    for b in range(B):   # per-episode phase (tau / delay may differ)
        rbf.pg.tau, rbf.pg.delay = tau[b], delay[b]
        basis = rbf.basis(times[b]) * tc.weights_scale
        wp = torch.nn.functional.pad(w[b], (zs, zg)) if zs + zg else w[b]
        pos[b] = torch.einsum('ik,jk->ij', basis, wp)
        if zs + zg:
            pos[b] += torch.as_tensor(init_pos[b], dtype=F)[None, :]
    vel = torch.zeros_like(pos)
    vel[:, :-1] = torch.diff(pos, dim=1) / torch.diff(times, dim=1)[..., None]
    vel[:, -1] = vel[:, -2]
    return pos.numpy(), vel.numpy()


def t_dmp(pc, bc, tc, params, duration, dt, init_time, init_pos, init_vel):
    tau, delay, wg = t_split(pc, params, tc.action_dim)
    B, D = wg.shape[0], tc.action_dim
    times = t_times(duration, dt, np.broadcast_to(init_time, (B,)))
    rbf = TRBF(TPhase(pc.phase_generator_type, pc.tau, pc.delay, pc.alpha_phase), bc.num_basis,
               bc.basis_bandwidth_factor, bc.num_basis_outside)
    T = times.shape[-1]
    pos, vel = torch.zeros(B, T, D), torch.zeros(B, T, D)
    alpha, beta = tc.alpha, tc.alpha / 4
    for b in range(B):
        rbf.pg.tau, rbf.pg.delay = tau[b], delay[b]
        w = wg[b, :, :-1] * tc.weights_scale
        g = wg[b, :, -1] * tc.goal_scale
        basis = rbf.basis(times[b])
        x = rbf.pg.phase(times[b])
        f = torch.einsum('ik,jk->ij', basis * x[..., None], w)
        pos[b, 0] = torch.as_tensor(init_pos[b], dtype=F)
        vel[b, 0] = torch.as_tensor(init_vel[b], dtype=F) * tau[b]
        sdt = torch.diff(rbf.pg.left_bound_linear_phase(times[b]), dim=-1)
        for i in range(T - 1):
            acc = alpha * (beta * (g - pos[b, i]) - vel[b, i]) + f[i]
            vel[b, i + 1] = vel[b, i] + sdt[i] * acc
            pos[b, i + 1] = pos[b, i] + sdt[i] * vel[b, i + 1]
        vel[b] = vel[b] / tau[b]
    return pos.numpy(), vel.numpy()


def t_prodmp(pc, bc, tc, params, duration, dt, init_time, init_pos, init_vel):
    tau, delay, local = t_split(pc, params, tc.action_dim)
    B, D, nb = local.shape[0], tc.action_dim, bc.num_basis
    it = torch.as_tensor(np.broadcast_to(init_time, (B,)).copy(), dtype=F)
    times = t_times(duration, dt, it)
    bg = TProDMPBasis(TPhase("exp", pc.tau, pc.delay, pc.alpha_phase), nb, bc.basis_bandwidth_factor, bc.alpha, bc.dt,
                      bc.pre_compute_length_factor)
    scale = torch.zeros(nb + 1)
    if tc.auto_scale_basis:
        scale[:] = bg.scale_factors
        scale[:-1] = scale[:-1] * tc.weights_scale
        scale[-1] = scale[-1] * tc.goal_scale
    else:
        scale[:-1], scale[-1] = tc.weights_scale, tc.goal_scale
    full = torch.zeros(B, D, nb + 1)
    c = 0
    if not tc.disable_weights:
        full[..., :nb] = local[..., :nb]; c = nb
    if not tc.disable_goal:
        full[..., nb] = local[..., c]
    ip, iv = torch.as_tensor(init_pos, dtype=F), torch.as_tensor(init_vel, dtype=F)
    if tc.relative_goal and tc.relative_goal_mode == "before_scale":
        full[..., -1] = full[..., -1] + ip            # init_pos joins the raw goal parameter (the default)
    wg = full * scale
    if tc.relative_goal and tc.relative_goal_mode == "after_scale":
        wg[..., -1] = wg[..., -1] + ip
    idx = bg.indices(times, tau, delay)
    idxb = bg.indices(it[:, None], tau, delay)[:, 0]
    y1, y2, dy1, dy2 = bg.y1[idx], bg.y2[idx], bg.dy1[idx], bg.dy2[idx]
    y1b, y2b, dy1b, dy2b = bg.y1[idxb], bg.y2[idxb], bg.dy1[idxb], bg.dy2[idxb]
    det = y1b * dy2b - y2b * dy1b
    xi1 = torch.einsum("...,...i->...i", dy2b / det, y1) - torch.einsum("...,...i->...i", dy1b / det, y2)
    xi2 = torch.einsum("...,...i->...i", y1b / det, y2) - torch.einsum("...,...i->...i", y2b / det, y1)
    xi3 = torch.einsum("...,...i->...i", dy2b / det, dy1) - torch.einsum("...,...i->...i", dy1b / det, dy2)
    xi4 = torch.einsum("...,...i->...i", y1b / det, dy2) - torch.einsum("...,...i->...i", y2b / det, dy1)
    pbi, vbi = bg.pc_pos[idxb], bg.pc_vel[idxb]
    pos_H = -(torch.einsum('...i,...j->...ij', xi1, pbi) + torch.einsum('...i,...j->...ij', xi2, vbi)) + bg.pc_pos[idx]
    vel_H = -(torch.einsum('...i,...j->...ij', xi3, pbi) + torch.einsum('...i,...j->...ij', xi4, vbi)) + bg.pc_vel[idx]
    vb = iv * tau[:, None]
    pos = torch.einsum('...i,...j->...ij', xi1, ip) + torch.einsum('...i,...j->...ij', xi2, vb) + \
        torch.einsum('...jk,...ik->...ji', pos_H, wg)
    vel = torch.einsum('...i,...j->...ij', xi3, ip) + torch.einsum('...i,...j->...ij', xi4, vb) + \
        torch.einsum('...jk,...ik->...ji', vel_H, wg)
    vel = vel / tau[:, None, None]
    return pos.numpy(), vel.numpy(), idx.numpy().astype(np.int32), idxb.numpy().astype(np.int32)


# ======================================================================================================================
# configurations
# ======================================================================================================================
PG = 0.01 * np.array([120., 120., 120., 120., 50., 30., 10.])
DG = 0.01 * np.array([10., 10., 10., 10., 6., 5., 3.])
TT_P = 0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0])
TT_D = 0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1])

CONFIGS = {
    "cfg1_promp_reacher5d": dict(
        pc=O.PhaseCfg("linear", tau=4.0),
        bc=O.BasisCfg("zero_rbf", num_basis=5, num_basis_zero_start=1, num_basis_zero_goal=0, basis_bandwidth_factor=3),
        tc=O.TrajCfg("promp", action_dim=5), dt=0.02, duration=4.0, B=2, init_times=[0.0], gains=(1.0, 0.1),
        act=(-1.0, 1.0)),
    "cfg2_prodmp_boxpushing": dict(
        pc=O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0),
        bc=O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=2, alpha=10),
        tc=O.TrajCfg("prodmp", action_dim=7), dt=0.02, duration=2.0, B=2, init_times=[0.0], gains=(PG, DG),
        act=(-1.0, 1.0)),
    "cfg3_dmp_reacher7d": dict(
        pc=O.PhaseCfg("exp", tau=4.0, alpha_phase=2.0),
        bc=O.BasisCfg("rbf", num_basis=5, basis_bandwidth_factor=3),
        tc=O.TrajCfg("dmp", action_dim=7, alpha=25.0), dt=0.02, duration=4.0, B=2, init_times=[0.0], gains=(1.0, 0.1),
        act=(-1.0, 1.0)),
    "cfg4_prodmp_replan": dict(
        pc=O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0),
        bc=O.BasisCfg("prodmp", num_basis=5, basis_bandwidth_factor=3, alpha=10),
        tc=O.TrajCfg("prodmp", action_dim=7, weights_scale=0.3, goal_scale=0.3, auto_scale_basis=True,
                     disable_goal=True),
        dt=0.02, duration=2.0, B=2, init_times=[0.0, 0.5, 1.0, 1.5], gains=(PG, DG), act=(-1.0, 1.0)),
    "cfg5_promp_tabletennis": dict(
        pc=O.PhaseCfg("linear", tau=2.8),
        bc=O.BasisCfg("zero_rbf", num_basis=3, num_basis_zero_start=1, num_basis_zero_goal=1, basis_bandwidth_factor=3),
        tc=O.TrajCfg("promp", action_dim=7), dt=0.008, duration=2.8, B=2, init_times=[0.0], gains=(TT_P, TT_D),
        act=(-1.0, 1.0)),
    # (tau = 1.5: _BB_DEFAULTS['ProDMP'] -- the TableTennis mp_config does not override it; rounds 1 - 5 had typed 2.8 = the episode's
    # duration here, found by the reference-generated config fixture tests/golden/ref_configs.json in round 6)
    "tt_prodmp_learn_tau_delay": dict(
        pc=O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0, learn_tau=True, learn_delay=True, tau_bound=(0.8, 1.5),
                      delay_bound=(0.05, 0.15)),
        bc=O.BasisCfg("prodmp", num_basis=3, basis_bandwidth_factor=3, alpha=25),
        tc=O.TrajCfg("prodmp", action_dim=7, weights_scale=0.7, auto_scale_basis=True, relative_goal=True,
                     disable_goal=True),
        dt=0.008, duration=2.8, B=2, init_times=[0.0], gains=(TT_P, TT_D), act=(-1.0, 1.0)),
}


def make_inputs(pc, bc, tc, B, seed):
    rng = np.random.default_rng(seed)
    P = O.num_params(pc, bc, tc)
    params = rng.standard_normal((B, P)).astype(np.float32)
    i = 0
    if pc.learn_tau:
        params[:, i] = rng.uniform(pc.tau_bound[0], pc.tau_bound[1], B); i += 1
    if pc.learn_delay:
        params[:, i] = rng.uniform(pc.delay_bound[0], pc.delay_bound[1], B); i += 1
    ip = rng.uniform(-1, 1, (B, tc.action_dim)).astype(np.float32)
    iv = rng.uniform(-0.5, 0.5, (B, tc.action_dim)).astype(np.float32)
    return params, ip, iv


def main():
    report = []
    for name, cfg in CONFIGS.items():
        pc, bc, tc, dt, duration, B = cfg["pc"], cfg["bc"], cfg["tc"], cfg["dt"], cfg["duration"], cfg["B"]
        params, ip, iv = make_inputs(pc, bc, tc, B, seed=abs(hash(name)) % 1000 if False else len(name))
        out = dict(params=params, init_pos=ip, init_vel=iv, init_times=np.array(cfg["init_times"], np.float64),
                   meta=np.array("generated by tests/golden/make_golden.py from the restatement in oracle/mp_oracle.py "
                                 "and an independent torch-CPU fp32 formulation; NOT from mp_pytorch (unavailable)"),
                   # provenance: the library versions the vectors were produced with and the exact generator script
                   versions=np.array(f"numpy {np.__version__}; torch {torch.__version__}; python {sys.version.split()[0]}"),
                   generator_sha256=np.array(GENERATOR_SHA256), oracle_sha256=np.array(ORACLE_SHA256))
        for k, it in enumerate(cfg["init_times"]):
            p32, v32 = O.get_trajectory(pc, bc, tc, params, duration, dt, it, ip, iv, dtype=np.float32)
            p64, v64 = O.get_trajectory(pc, bc, tc, params, duration, dt, it, ip, iv, dtype=np.float64)
            t = tc.trajectory_generator_type
            if t == "promp":
                tp, tv = t_promp(pc, bc, tc, params, duration, dt, it, ip)
            elif t == "dmp":
                tp, tv = t_dmp(pc, bc, tc, params, duration, dt, it, ip, iv)
            else:
                tp, tv, tidx, tidxb = t_prodmp(pc, bc, tc, params, duration, dt, it, ip, iv)
                _, _, oidx, oidxb = O.prodmp_trajectory(pc, bc, tc, np.clip(params, *O.params_bounds(pc, bc, tc)),
                                                        O.make_times(duration, dt, it), it, ip, iv,
                                                        dtype=np.float32, return_indices=True)
                assert np.array_equal(oidx.astype(np.int32), tidx), f"{name}: table indices differ between formulations"
                assert np.array_equal(oidxb.astype(np.int32), tidxb)
                out[f"idx_{k}"], out[f"idxb_{k}"] = tidx, tidxb
            if not (pc.learn_tau or pc.learn_delay):
                pass
            sp, sv = np.abs(p64).max(), np.abs(v64).max()
            report.append((name, it, np.abs(p32 - p64).max() / sp, np.abs(v32 - v64).max() / sv,
                           np.abs(tp - p64).max() / sp, np.abs(tv - v64).max() / sv))
            out[f"pos32_{k}"], out[f"vel32_{k}"] = p32, v32
            out[f"tpos_{k}"], out[f"tvel_{k}"] = tp.astype(np.float32), tv.astype(np.float32)
            out[f"err64_{k}"] = np.array([np.abs(p32 - p64).max() / sp, np.abs(v32 - v64).max() / sv])
            if k == 0:
                pg, dg = cfg["gains"]
                lo, hi = cfg["act"]
                for plant in ("static", "double_integrator"):
                    a, q, qd = O.rollout(p32, v32, "motor", pg, dg, lo, hi, plant, dt, ip.astype(np.float64),
                                         iv.astype(np.float64))
                    # actions as the fp32 the device path stores; final plant state in full float64
                    out[f"act_{plant}"], out[f"q_{plant}"], out[f"qd_{plant}"] = a.astype(np.float32), q, qd
        # torch.linspace (machine-dependent vectorised kernel) vs the oracle's scalar recipe
        tl = torch.linspace(0, duration, round(duration / dt) + 1, dtype=F).numpy()[1:]
        ol = O.make_times(duration, dt, 0.0)
        out["times_torch"], out["times_oracle"] = tl, ol
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(f"{name}: wrote {os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.1f} KB; "
              f"linspace max |torch - oracle| = {np.abs(tl - ol).max():.2e}")
    print("\nrelative to max|oracle64| per array:")
    print(f"{'config':30s} {'t0':>5s} {'o32 pos':>9s} {'o32 vel':>9s} {'torch pos':>9s} {'torch vel':>9s}")
    for r in report:
        print(f"{r[0]:30s} {r[1]:5.2f} {r[2]:9.2e} {r[3]:9.2e} {r[4]:9.2e} {r[5]:9.2e}")


if __name__ == "__main__":
    main()
