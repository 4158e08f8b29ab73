"""facade: mp_pytorch.mp -- ProMP / DMP / ProDMP with the MPInterface surface BlackBoxWrapper touches
(black_box_wrapper.py:57,62-65,102,106,113-118,124-125,226), evaluated by the oracle one episode at a time"""
import numpy as np
import torch

from oracle import mp_oracle as O
from . import BEHAVIOUR


class _MP:
    kind = None

    def __init__(self, basis_gn, num_dof, weights_scale=1.0, **kwargs):
        ph = basis_gn.phase_generator
        self.phase_gn = ph
        self.pc = O.PhaseCfg(ph.kind, tau=ph.tau, delay=ph.delay, alpha_phase=ph.alpha_phase, learn_tau=ph.learn_tau,
                             learn_delay=ph.learn_delay, tau_bound=tuple(ph.tau_bound), delay_bound=tuple(ph.delay_bound))
        self.bc = O.BasisCfg(basis_gn.kind, single_rbf_mode=BEHAVIOUR["single_rbf_mode"], **basis_gn.kw)
        known = {f for f in ("goal_scale", "alpha", "auto_scale_basis", "relative_goal", "disable_goal", "disable_weights")
                 if f in kwargs}
        tk = {k: kwargs[k] for k in known}
        self.tc = O.TrajCfg(self.kind, action_dim=int(num_dof), weights_scale=float(weights_scale),
                            relative_goal_mode=BEHAVIOUR["relative_goal_mode"],
                            goal_offset_mode=BEHAVIOUR["goal_offset_mode"],
                            goal_offset=float(kwargs.get("goal_offset", 0.0)),
                            dmp_first_sample=BEHAVIOUR["dmp_first_sample"], **tk)
        O.rbf_centers_bandwidth(self.pc, self.bc) if self.bc.basis_generator_type != "prodmp" else None   # may refuse
        self.learn_tau, self.tau = ph.learn_tau, ph.tau
        self.reset()

    def reset(self):
        self.params = self.init = None
        self.duration, self.dt = None, None

    def set_duration(self, duration, dt):
        self.duration, self.dt = duration, float(dt)

    def set_params(self, params):
        self.params = np.asarray(params, np.float32)

    def set_initial_conditions(self, init_time, init_pos, init_vel):
        self.init = (float(np.asarray(init_time)), np.asarray(init_pos, np.float32), np.asarray(init_vel, np.float32))

    def get_params_bounds(self):
        b = O.params_bounds(self.pc, self.bc, self.tc)
        return torch.from_numpy(b[0].copy()), torch.from_numpy(b[1].copy())

    def _traj(self):
        it, ip, iv = self.init
        return O.get_trajectory(self.pc, self.bc, self.tc, self.params[None], self.duration, self.dt, it, ip[None], iv[None],
                                dtype=np.float32, clip=False)

    def get_traj_pos(self):
        return torch.from_numpy(np.ascontiguousarray(self._traj()[0][0]))

    def get_traj_vel(self):
        return torch.from_numpy(np.ascontiguousarray(self._traj()[1][0]))


class ProMP(_MP):
    kind = "promp"


class DMP(_MP):
    kind = "dmp"


class ProDMP(_MP):
    kind = "prodmp"
