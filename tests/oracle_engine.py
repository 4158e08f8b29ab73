"""
An object with the few TrajectoryEngine methods tests/test_gpu_ode.py's checks use, answering from the CPU oracle
(float64) instead of the GPU: lets the SAME differential-equation checks that the GPU suite applies to the HIP path run
against the oracle in the CPU suite.  Construction-time constants still come from libmpk's device-free host views
(``mpk_host_rbf``, ``mpk_host_prodmp_tables``), so both suites solve the same ODE.
"""
import ctypes as C

import numpy as np
import torch

from fancy_gym_amd import _lib
from oracle import mp_oracle as O


class OracleEngine:
    device = torch.device("cpu")

    def __init__(self, pc, bc, tc, dt, duration):
        self.pc, self.bc, self.tc, self.dt, self.duration = pc, bc, tc, dt, duration
        c = _lib.mpk_config()
        c.abi_version = _lib.MPK_ABI_VERSION
        c.mp_type = _lib.MP_TYPES[tc.trajectory_generator_type]
        c.phase_type = _lib.PHASE_TYPES[pc.phase_generator_type]
        c.basis_type = _lib.BASIS_TYPES[bc.basis_generator_type]
        c.num_dof, c.num_basis, c.num_basis_outside = tc.action_dim, bc.num_basis, bc.num_basis_outside
        if bc.basis_generator_type == "zero_rbf":
            c.num_basis_zero_start, c.num_basis_zero_goal = bc.num_basis_zero_start, bc.num_basis_zero_goal
        c.learn_tau, c.learn_delay = int(pc.learn_tau), int(pc.learn_delay)
        c.auto_scale_basis, c.relative_goal = int(tc.auto_scale_basis), int(tc.relative_goal)
        c.disable_goal, c.disable_weights = int(tc.disable_goal), int(tc.disable_weights)
        c.pre_compute_length_factor = bc.pre_compute_length_factor
        c.tau, c.delay, c.alpha_phase = pc.tau, pc.delay, pc.alpha_phase
        c.tau_bound[0], c.tau_bound[1] = pc.tau_bound
        c.delay_bound[0], c.delay_bound[1] = pc.delay_bound
        c.basis_bandwidth_factor, c.basis_alpha, c.basis_dt = bc.basis_bandwidth_factor, bc.alpha, bc.dt
        c.weights_scale, c.goal_scale, c.dmp_alpha = tc.weights_scale, tc.goal_scale, tc.alpha
        c.dt, c.duration = dt, duration
        self.config = c

    def trajectory(self, params, init_pos, init_vel, init_time=0.0):
        if isinstance(init_time, torch.Tensor):
            init_time = init_time.cpu().numpy()
        p, v = O.get_trajectory(self.pc, self.bc, self.tc, params, self.duration, self.dt, init_time, init_pos,
                                init_vel, dtype=np.float64)
        return torch.from_numpy(p), torch.from_numpy(v)

    def check_range(self):
        pass

    def times(self):
        return O.make_times(self.duration, self.dt, 0.0)

    def prodmp_tables(self):
        lib = _lib.load()
        n = lib.mpk_host_prodmp_tables(C.byref(self.config), *[None] * 8)
        assert n > 0, _lib.last_error()
        K = self.bc.num_basis + 1
        arrs = [np.empty(n) for _ in range(4)] + [np.empty((n, K)) for _ in range(2)] + [np.empty(K)]
        sd = C.c_float()
        assert lib.mpk_host_prodmp_tables(C.byref(self.config), *[a.ctypes.data for a in arrs], C.byref(sd)) == n
        return dict(zip(("y1", "y2", "dy1", "dy2", "pos_basis", "vel_basis", "scale"), arrs))
