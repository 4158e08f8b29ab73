"""
TrajectoryEngine -- thin, typed wrapper over one ``mpk_handle`` (include/mpk.h).

It owns no arithmetic: every number it returns was produced by the HIP kernels in ``csrc/``.  PyTorch is used only for
device memory and streams (tensors are passed to the C-ABI as raw device pointers).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import (BASIS_TYPES, CTRL_TYPES, MP_TYPES, PHASE_TYPES, PLANT_TYPES, MPKLibraryError, mpk_config,
                   mpk_rollout_cfg)

_INF = float("inf")


def _dptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dvec(x, n: int):
    """scalar or sequence -> (ctypes double[n], keep-alive)"""
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)))
    return a.ctypes.data_as(C.POINTER(C.c_double)), a


class RolloutSpec:
    """Controller + plant description for the device rollout (mpk_rollout_cfg)."""

    def __init__(self, controller_type: str, num_dof: int, p_gains=1.0, d_gains=0.5, act_low=-_INF, act_high=_INF,
                 plant: str = "static", dt: float = 0.0):
        controller_type = controller_type.lower()
        # 'metaworld' (controller/meta_world_controller.py:15-25: xyz as a position delta, last entry = the raw gripper
        # opening) is the motor controller with unit position gains, no velocity gains and a zero in the last entry of the
        # current position -- 1.0 * (des - cur) + 0.0 * (..) is exact in float64 -- so it runs on the motor kernels; there is
        # no batchable metaworld plant, hence open loop (frozen state) only: TrajectoryEngine.trajectory_actions / pd_rollout
        self.metaworld = controller_type == "metaworld"
        if self.metaworld:
            if plant != "static":
                raise ValueError("the metaworld controller has no device plant: use plant='static' (actions for a frozen "
                                 "state) or step the environments on the host (VectorBlackBox)")
            controller_type, p_gains, d_gains = "motor", 1.0, 0.0
        if controller_type not in CTRL_TYPES:
            raise ValueError(f"controller type {controller_type!r} has no device implementation; "
                             f"choose one of {list(CTRL_TYPES)}")
        if plant not in PLANT_TYPES:
            raise ValueError(f"unknown plant {plant!r}; choose one of {list(PLANT_TYPES)}")
        self.controller_type, self.plant, self.num_dof, self.dt = controller_type, plant, num_dof, float(dt)
        self._keep = []
        self.c = mpk_rollout_cfg()
        self.c.controller_type = CTRL_TYPES[controller_type]
        self.c.plant_type = PLANT_TYPES[plant]
        self.c.dt = float(dt)
        for name, val in (("p_gains", p_gains), ("d_gains", d_gains), ("act_low", act_low), ("act_high", act_high)):
            v = np.asarray(val, dtype=np.float64)
            if v.ndim > 0 and v.shape != (num_dof,):
                raise ValueError(f"Mismatch in dimension between {name} {v.shape} and the action dimension ({num_dof},)")
            ptr, keep = _dvec(val, num_dof)
            setattr(self.c, name, ptr)
            self._keep.append(keep)


class TrajectoryEngine:
    """
    One configured movement primitive on one GPU.  Constructor arguments mirror the kwarg groups the reference hands
    to mp_pytorch (fancy_gym/utils/make_env_helpers.py:128-131; fancy_gym/envs/registry.py:62-129).
    """

    def __init__(self, mp_type: str, phase_type: str, basis_type: str, num_dof: int, num_basis: int, *, dt: float,
                 duration: float, tau: float, delay: float = 0.0, alpha_phase: float = 3.0, learn_tau: bool = False,
                 learn_delay: bool = False, tau_bound: Sequence[float] = (1e-5, _INF),
                 delay_bound: Sequence[float] = (0.0, _INF), basis_bandwidth_factor: float = 3.0,
                 num_basis_outside: int = 0, num_basis_zero_start: int = 0, num_basis_zero_goal: int = 0,
                 basis_alpha: float = 25.0, basis_dt: float = 0.01, pre_compute_length_factor: int = 6,
                 weights_scale: float = 1.0, goal_scale: float = 1.0, dmp_alpha: float = 25.0,
                 auto_scale_basis: bool = False, relative_goal: bool = False, disable_goal: bool = False,
                 disable_weights: bool = False, relative_goal_mode: str = "before_scale",
                 goal_offset_mode: str = "ignore", goal_offset: float = 0.0, single_rbf_mode: str = "unit_gap",
                 dmp_first_sample: str = "init", device: Union[int, torch.device, None] = None):
        self._h = C.c_void_p()
        self._lib = _lib.load()
        for name, table, val in (("movement primitive", MP_TYPES, mp_type), ("phase generator", PHASE_TYPES, phase_type),
                                 ("basis generator", BASIS_TYPES, basis_type)):
            if val not in table:
                raise ValueError(f"Specified {name} type {val} not supported, please choose one of {list(table)}.")
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if isinstance(device, torch.device):
            device = device.index or 0
        c = mpk_config()
        c.abi_version = _lib.MPK_ABI_VERSION
        c.device = int(device)
        c.mp_type, c.phase_type, c.basis_type = MP_TYPES[mp_type], PHASE_TYPES[phase_type], BASIS_TYPES[basis_type]
        c.num_dof, c.num_basis = int(num_dof), int(num_basis)
        c.num_basis_outside = int(num_basis_outside)
        c.num_basis_zero_start, c.num_basis_zero_goal = int(num_basis_zero_start), int(num_basis_zero_goal)
        c.learn_tau, c.learn_delay = int(bool(learn_tau)), int(bool(learn_delay))
        c.auto_scale_basis, c.relative_goal = int(bool(auto_scale_basis)), int(bool(relative_goal))
        c.disable_goal, c.disable_weights = int(bool(disable_goal)), int(bool(disable_weights))
        c.pre_compute_length_factor = int(pre_compute_length_factor)
        # the "(?)" items of SURVEY Appendix A as named options (include/mpk.h MPK_RELGOAL_* ...); first = default
        for field, table, val in (("relative_goal_mode", _lib.RELATIVE_GOAL_MODES, relative_goal_mode),
                                  ("goal_offset_mode", _lib.GOAL_OFFSET_MODES, goal_offset_mode),
                                  ("single_rbf_mode", _lib.SINGLE_RBF_MODES, single_rbf_mode),
                                  ("dmp_first_sample", _lib.DMP_FIRST_SAMPLE_MODES, dmp_first_sample)):
            if val not in table:
                raise ValueError(f"{field} {val!r} not supported, please choose one of {list(table)}.")
            setattr(c, field, table[val])
        c.goal_offset = float(goal_offset or 0.0)
        c.tau, c.delay, c.alpha_phase = float(tau), float(delay), float(alpha_phase)
        c.tau_bound[0], c.tau_bound[1] = float(tau_bound[0]), float(tau_bound[1])
        c.delay_bound[0], c.delay_bound[1] = float(delay_bound[0]), float(delay_bound[1])
        c.basis_bandwidth_factor, c.basis_alpha, c.basis_dt = float(basis_bandwidth_factor), float(basis_alpha), float(basis_dt)
        c.weights_scale, c.goal_scale, c.dmp_alpha = float(weights_scale), float(goal_scale), float(dmp_alpha)
        c.dt, c.duration = float(dt), float(duration)
        self.config = c
        self.mp_type, self.phase_type, self.basis_type = mp_type, phase_type, basis_type
        self.device = torch.device("cuda", int(device))
        _lib.check(self._lib.mpk_create(C.byref(c), C.byref(self._h)))
        self.num_dof = self._lib.mpk_num_dof(self._h)
        self.num_params = self._lib.mpk_num_params(self._h)
        self._host_stage = None     # pinned / device staging of trajectory_host, allocated on first use
        # B = 1 host path: the kernel reads / writes the pinned host buffers directly (False: staged device copies)
        self.b1_zero_copy = os.environ.get("MPK_B1_ZEROCOPY", "1") != "0"

    # ---- lifecycle ---------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mpk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- shape / bounds ----------------------------------------------------------------------------------------
    @property
    def num_steps(self) -> int:
        return self._lib.mpk_num_steps(self._h)

    def params_bounds(self) -> np.ndarray:
        """[2, P] float32 (traj_gen.get_params_bounds(), black_box_wrapper.py:122-127)."""
        out = np.empty((2, self.num_params), np.float32)
        _lib.check(self._lib.mpk_params_bounds(self._h, out[0].ctypes.data, out[1].ctypes.data))
        return out

    def set_duration(self, duration: float, dt: float):
        _lib.check(self._lib.mpk_set_duration(self._h, float(duration), float(dt)))

    def times(self) -> np.ndarray:
        t = np.empty(self.num_steps, np.float32)
        _lib.check(self._lib.mpk_times(self._h, t.ctypes.data))
        return t

    # ---- helpers -----------------------------------------------------------------------------------------------
    def _f32(self, x, shape) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        if t.shape != shape:
            t = t.expand(shape)
        return t.contiguous()

    def _f64(self, x, shape) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        if t.shape != shape:
            t = t.expand(shape)
        return t.contiguous()

    def _stream(self) -> int:
        """the current stream of THIS engine's device (not of whatever device is current)"""
        return torch.cuda.current_stream(self.device).cuda_stream

    def set_option(self, key: str, value: int = _lib.MPK_OPT_AUTO):
        """kernel-selection override for this engine only (mpk_set_option; A/B runs and variant-pinning tests)"""
        _lib.set_option(key, value, self._h)

    def get_option(self, key: str) -> int:
        return _lib.get_option(key, self._h)

    # ---- hot path ----------------------------------------------------------------------------------------------
    def trajectory(self, params, init_pos, init_vel, init_time=0.0, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        params [B, P] -> (pos [B, T, D], vel [B, T, D]) float32 CUDA tensors.  ``init_time`` is a python float shared by
        all episodes, or a tensor [B] of per-episode values.
        """
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.dim() == 1:
            params = params[None]
        params = params.contiguous()
        B = params.shape[0]
        if params.shape[1] != self.num_params:
            raise ValueError(f"params has {params.shape[1]} entries per episode, expected {self.num_params}")
        D, T = self.num_dof, self.num_steps
        init_pos, init_vel = self._f32(init_pos, (B, D)), self._f32(init_vel, (B, D))
        it_t = None
        it_s = 0.0
        if isinstance(init_time, torch.Tensor) and init_time.dim() > 0:
            it_t = self._f32(init_time, (B,))
        else:
            it_s = float(init_time)
        if out is None:
            pos = torch.empty((B, T, D), dtype=torch.float32, device=self.device)
            vel = torch.empty((B, T, D), dtype=torch.float32, device=self.device)
        else:
            pos, vel = out
        _lib.check(self._lib.mpk_trajectory(self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(),
                                            _dptr(it_t), it_s, pos.data_ptr(), vel.data_ptr(), B, self._stream()))
        return pos, vel

    def check_range(self):
        """synchronise and raise RuntimeError if a per-episode-phase ProDMP launch left the pre-computed table range"""
        _lib.check(self._lib.mpk_check_range(self._h, self._stream()))

    def poll_fault(self):
        """raise MPKLibraryError if a FINISHED launch of the ring kernels gave up waiting (outputs incomplete); synchronises nothing
        (mpk.h: mpk_poll_fault) -- for callers that synchronise themselves and make no further call that would report it"""
        _lib.check(self._lib.mpk_poll_fault(self._h))

    def gate_flags(self, valid: torch.Tensor, was_done: Optional[torch.Tensor], done: torch.Tensor):
        """(terminated, truncated) of a gated step as bool [B] tensors in ONE launch (mpk.h: mpk_gate_flags): an invalid plan terminates
        an episode that was live, a valid one that finished is truncated (black_box_wrapper.py:169-172,198-203); uint8 / bool inputs"""
        B = valid.shape[0]
        out = torch.empty((2, B), dtype=torch.uint8, device=self.device)
        u8 = lambda t: None if t is None else (t if t.dtype == torch.uint8 else t.view(torch.uint8))      # noqa: E731
        v, w, d = u8(valid), u8(was_done), u8(done)
        _lib.check(self._lib.mpk_gate_flags(self._h, v.data_ptr(), _dptr(w), d.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), B,
                                            self._stream()))
        return out[0].view(torch.bool), out[1].view(torch.bool)

    def trajectory_host(self, params: np.ndarray, init_pos: np.ndarray, init_vel: np.ndarray, init_time: float
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
        """
        ONE episode, host arrays in, host tensors out (what BlackBoxWrapper.get_trajectory needs at B = 1): inputs travel
        in one pinned buffer, (pos | vel) come back in one, a single stream synchronisation.  Returns CPU float32
        tensors [T, D] -- like mp_pytorch on its default device, so fancy_gym's get_numpy does not copy again.
        """
        P, D, T = self.num_params, self.num_dof, self.num_steps
        st = self._host_stage
        if st is None or st[0] != (P, D, T):
            n_in = P + 2 * D + 1
            st = ((P, D, T), torch.empty(n_in, dtype=torch.float32).pin_memory(),
                  torch.empty(n_in, dtype=torch.float32, device=self.device),
                  torch.empty((2, T, D), dtype=torch.float32, device=self.device),
                  torch.empty((2, T, D), dtype=torch.float32).pin_memory())
            self._host_stage = st
        _, h_in, d_in, d_out, h_out = st
        buf = h_in.numpy()
        buf[:P] = np.asarray(params, np.float32).reshape(-1)
        buf[P:P + D] = np.asarray(init_pos, np.float32).reshape(-1)
        buf[P + D:P + 2 * D] = np.asarray(init_vel, np.float32).reshape(-1)
        buf[P + 2 * D] = init_time
        per_episode = bool(self.config.learn_tau or self.config.learn_delay)
        # zero copy (default): the kernel reads the inputs from, and writes (pos | vel) to, the pinned host buffers
        # themselves -- pinned host memory is device-accessible at the same address, and a few KB over PCIe cost less than
        # the two copy calls (MPK_B1_ZEROCOPY=0: staged copies through device buffers instead)
        zero_copy = self.b1_zero_copy
        if not zero_copy:
            d_in.copy_(h_in, non_blocking=True)
        base = (h_in if zero_copy else d_in).data_ptr()
        dst = h_out if zero_copy else d_out
        _lib.check(self._lib.mpk_trajectory(self._h, base, base + 4 * P, base + 4 * (P + D),
                                            base + 4 * (P + 2 * D) if per_episode else None, float(init_time),
                                            dst[0].data_ptr(), dst[1].data_ptr(), 1, self._stream()))
        if not zero_copy:
            h_out.copy_(d_out, non_blocking=True)
        if per_episode and self.mp_type == "prodmp":
            self.check_range()                       # synchronises
        else:
            torch.cuda.current_stream(self.device).synchronize()
        out = h_out.clone()
        return out[0], out[1]

    @staticmethod
    def _metaworld_state(c_pos: torch.Tensor):
        """current state as the motor kernels must see it for the metaworld controller: the gripper entry of the
        position is not subtracted (0), velocities play no part"""
        c_pos = c_pos.clone()
        c_pos[:, -1] = 0.0
        return c_pos, torch.zeros_like(c_pos)

    @staticmethod
    def _refuse_metaworld(spec, entry: str):
        """entry points that update the plant state in place cannot zero the gripper entry of the state the way
        _metaworld_state does for a frozen one: running the plain motor law there would turn the gripper action into
        des - q[-1] without a word"""
        if getattr(spec, "metaworld", False):
            raise ValueError(f"{entry}: the metaworld controller runs on the device for a frozen state only "
                             "(trajectory_actions / pd_rollout with plant='static'); step metaworld environments on the "
                             "host (VectorBlackBox)")

    def trajectory_actions(self, params, init_pos, init_vel, spec: RolloutSpec, c_pos, c_vel, init_time: float = 0.0,
                           out=None):
        """Fused trajectory + open-loop controller actions for a state frozen over the plan (MPK_PLANT_STATIC)."""
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.dim() == 1:
            params = params[None]
        params = params.contiguous()
        B, D, T = params.shape[0], self.num_dof, self.num_steps
        init_pos, init_vel = self._f32(init_pos, (B, D)), self._f32(init_vel, (B, D))
        c_pos, c_vel = self._f64(c_pos, (B, D)), self._f64(c_vel, (B, D))
        if getattr(spec, "metaworld", False):
            c_pos, c_vel = self._metaworld_state(c_pos)
        if out is None:
            pos, vel, act = (torch.empty((B, T, D), dtype=torch.float32, device=self.device) for _ in range(3))
        else:
            pos, vel, act = out
        _lib.check(self._lib.mpk_trajectory_actions(
            self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
            c_pos.data_ptr(), c_vel.data_ptr(), pos.data_ptr(), vel.data_ptr(), act.data_ptr(), B, self._stream()))
        return pos, vel, act

    def trajectory_rollout(self, params, init_pos, init_vel, spec: RolloutSpec, q: torch.Tensor, qd: torch.Tensor,
                           n_steps: Optional[torch.Tensor] = None, init_time: float = 0.0, out=None):
        """
        One fused launch: trajectory + closed-loop controller / double-integrator rollout (BlackBoxWrapper.step for a
        GPU-resident plant).  q, qd float64 [B, D] are updated in place.  Returns (pos, vel, actions).
        """
        self._refuse_metaworld(spec, "trajectory_rollout")
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.dim() == 1:
            params = params[None]
        params = params.contiguous()
        B, D, T = params.shape[0], self.num_dof, self.num_steps
        init_pos, init_vel = self._f32(init_pos, (B, D)), self._f32(init_vel, (B, D))
        assert q.dtype == torch.float64 and qd.dtype == torch.float64 and q.is_contiguous() and qd.is_contiguous()
        if out is None:
            pos, vel, act = (torch.empty((B, T, D), dtype=torch.float32, device=self.device) for _ in range(3))
        else:
            pos, vel, act = out
        if n_steps is not None:
            n_steps = n_steps.to(device=self.device, dtype=torch.int32).contiguous()
        _lib.check(self._lib.mpk_trajectory_rollout(
            self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
            q.data_ptr(), qd.data_ptr(), _dptr(n_steps), pos.data_ptr(), vel.data_ptr(), act.data_ptr(), B,
            self._stream()))
        return pos, vel, act

    def episode_reset(self, q: torch.Tensor, qd: torch.Tensor, traj_steps: torch.Tensor, plan_steps: torch.Tensor,
                      done: torch.Tensor, init_q: Optional[torch.Tensor] = None, init_qd: Optional[torch.Tensor] = None,
                      cond: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """BlackBoxWrapper.reset for device-resident episodes in one launch (mpk.h: mpk_episode_reset); everything in place"""
        B, D = q.shape
        for t, dt_ in ((q, torch.float64), (qd, torch.float64), (traj_steps, torch.int32), (plan_steps, torch.int32),
                       (done, torch.uint8)):
            assert t.dtype == dt_ and t.is_contiguous()
        for t in (init_q, init_qd):
            assert t is None or (t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (B, D))
        cp, cv = cond if cond is not None else (None, None)
        _lib.check(self._lib.mpk_episode_reset(self._h, _dptr(init_q), _dptr(init_qd), q.data_ptr(), qd.data_ptr(),
                                               _dptr(cp), _dptr(cv), traj_steps.data_ptr(), plan_steps.data_ptr(),
                                               done.data_ptr(), B, self._stream()))

    def _gate(self, gate, B: int):
        """
        dict(pos_low, pos_high, check_tau_delay=False, tau_bound=None, delay_bound=None, raw_params=None, valid=None, penalty=None)
        -> (mpk_validity_gate, valid uint8 [B], penalty float64 [B], keep-alive); mpk.h: mpk_validity_gate
        """
        D = self.num_dof
        lo_p, lo_k = _dvec(gate["pos_low"], D)
        hi_p, hi_k = _dvec(gate["pos_high"], D)
        check = bool(gate.get("check_tau_delay"))
        g = _lib.mpk_validity_gate()
        g.pos_low, g.pos_high = lo_p, hi_p
        g.check_tau_delay = int(check)
        tb = gate.get("tau_bound") if check else None
        db = gate.get("delay_bound") if check else None
        if check and (tb is None or db is None):
            raise ValueError("check_tau_delay needs tau_bound and delay_bound")
        g.tau_bound[0], g.tau_bound[1] = (float(tb[0]), float(tb[1])) if check else (0.0, 0.0)
        g.delay_bound[0], g.delay_bound[1] = (float(db[0]), float(db[1])) if check else (0.0, 0.0)
        raw = gate.get("raw_params")
        if raw is not None:
            raw = torch.as_tensor(raw, dtype=torch.float32, device=self.device).contiguous()
            if tuple(raw.shape) != (B, self.num_params):
                raise ValueError(f"raw_params must be [{B}, {self.num_params}], got {tuple(raw.shape)}")
        valid = gate.get("valid")
        if valid is None:
            valid = torch.empty(B, dtype=torch.uint8, device=self.device)
        pen = gate.get("penalty")
        if pen is None:
            pen = torch.empty(B, dtype=torch.float64, device=self.device)
        assert valid.dtype == torch.uint8 and pen.dtype == torch.float64 and valid.is_contiguous() and pen.is_contiguous()
        g.raw_params, g.valid, g.penalty = _dptr(raw), valid.data_ptr(), pen.data_ptr()
        return g, valid, pen, (lo_k, hi_k, raw)

    def replan_step(self, params, init_pos, init_vel, spec: RolloutSpec, q: torch.Tensor, qd: torch.Tensor,
                    traj_steps: torch.Tensor, plan_steps: torch.Tensor, done: torch.Tensor, every: int,
                    max_planning_times: int, horizon: int, init_time: float = 0.0, condition: bool = False, out=None,
                    gate=None):
        """
        One replanning step of BlackBoxWrapper.step for every episode (mpk.h: mpk_replan_step): integer state, plan,
        controller + plant for the executed steps and -- ``condition`` -- the desired state at the last executed step, in
        ONE launch where the fused closed-loop kernel applies.  q, qd, traj_steps, plan_steps, done are updated in place.
        Returns dict(pos, vel, actions, seg_len int32 [B], done uint8 [B] snapshot, cond_pos, cond_vel (or None)).
        ``gate`` (see ``_gate``): the validity gate inside the step (mpk.h: mpk_replan_step_gated) -- additionally ``valid``
        uint8 [B] and ``penalty`` float64 [B]; an invalid plan finishes its episode without a step.
        """
        self._refuse_metaworld(spec, "replan_step")
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.dim() == 1:
            params = params[None]
        params = params.contiguous()
        B, D, T = params.shape[0], self.num_dof, self.num_steps
        init_pos, init_vel = self._f32(init_pos, (B, D)), self._f32(init_vel, (B, D))
        assert q.dtype == torch.float64 and qd.dtype == torch.float64 and q.is_contiguous() and qd.is_contiguous()
        assert traj_steps.dtype == torch.int32 and plan_steps.dtype == torch.int32 and done.dtype == torch.uint8
        if out is None:
            pos, vel, act = (torch.empty((B, T, D), dtype=torch.float32, device=self.device) for _ in range(3))
        else:
            pos, vel, act = out
        seg = torch.empty(B, dtype=torch.int32, device=self.device)
        done_out = torch.empty(B, dtype=torch.uint8, device=self.device)
        cp = cv = None
        if condition:
            cp, cv = (torch.empty((B, D), dtype=torch.float32, device=self.device) for _ in range(2))
        st = _lib.mpk_replan_state(traj_steps.data_ptr(), plan_steps.data_ptr(), done.data_ptr(), seg.data_ptr(),
                                   done_out.data_ptr(), _dptr(cp), _dptr(cv), int(every),
                                   int(min(max_planning_times, 2 ** 31 - 1)), int(horizon), 0)
        if gate is not None:
            g, valid, pen, keep = self._gate(gate, B)
            _lib.check(self._lib.mpk_replan_step_gated(
                self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
                q.data_ptr(), qd.data_ptr(), C.byref(st), C.byref(g), pos.data_ptr(), vel.data_ptr(), act.data_ptr(), B,
                self._stream()))
            return dict(pos=pos, vel=vel, actions=act, seg_len=seg, done=done_out, cond_pos=cp, cond_vel=cv, valid=valid,
                        penalty=pen)
        _lib.check(self._lib.mpk_replan_step(
            self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
            q.data_ptr(), qd.data_ptr(), C.byref(st), pos.data_ptr(), vel.data_ptr(), act.data_ptr(), B,
            self._stream()))
        return dict(pos=pos, vel=vel, actions=act, seg_len=seg, done=done_out, cond_pos=cp, cond_vel=cv)

    def episode_return(self, params, init_pos, init_vel, spec: RolloutSpec, q: torch.Tensor, qd: torch.Tensor, *,
                       replan=None, n_steps: Optional[torch.Tensor] = None, reward: Optional[str] = None,
                       goal: Optional[torch.Tensor] = None, step0: Optional[torch.Tensor] = None,
                       steps_before_reward: int = 199, aggregation: str = "sum", init_time: float = 0.0,
                       condition: bool = False, gate=None):
        """
        One plan of a ``verbose < 2`` step for every episode in ONE launch, nothing per step stored (mpk.h: mpk_episode_return):
        plan + controller + double-integrator plant + reward + reward_aggregation (+ the integer replanning state and the
        condition gather when ``replan = (traj_steps, plan_steps, done, every, max_planning_times, horizon)``).  q, qd (and
        the replanning state) are updated in place.  Returns dict(ret float64 [B], seg_len int32 [B], done uint8 [B] or None,
        cond_pos, cond_vel (or None)).  NotImplementedError where the fused kernel does not apply (per-episode phase, > 16
        columns / DoF, very long horizons): use replan_step / trajectory_rollout + reacher_rollout + reward_aggregate.
        """
        self._refuse_metaworld(spec, "episode_return")
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.dim() == 1:
            params = params[None]
        params = params.contiguous()
        B, D = params.shape[0], self.num_dof
        init_pos, init_vel = self._f32(init_pos, (B, D)), self._f32(init_vel, (B, D))
        assert q.dtype == torch.float64 and qd.dtype == torch.float64 and q.is_contiguous() and qd.is_contiguous()
        ret = torch.empty(B, dtype=torch.float64, device=self.device)
        seg = torch.empty(B, dtype=torch.int32, device=self.device)
        done_out = cp = cv = None
        st = None
        if replan is not None:
            traj_steps, plan_steps, done, every, max_planning_times, horizon = replan
            assert traj_steps.dtype == torch.int32 and plan_steps.dtype == torch.int32 and done.dtype == torch.uint8
            done_out = torch.empty(B, dtype=torch.uint8, device=self.device)
            if condition:
                cp, cv = (torch.empty((B, D), dtype=torch.float32, device=self.device) for _ in range(2))
            st = _lib.mpk_replan_state(traj_steps.data_ptr(), plan_steps.data_ptr(), done.data_ptr(), seg.data_ptr(),
                                       done_out.data_ptr(), _dptr(cp), _dptr(cv), int(every),
                                       int(min(max_planning_times, 2 ** 31 - 1)), int(horizon), 0)
        elif n_steps is not None:
            n_steps = n_steps.to(device=self.device, dtype=torch.int32).contiguous()
        if goal is not None:
            goal = torch.as_tensor(goal, dtype=torch.float64, device=self.device).expand(B, 2).contiguous()
        if step0 is not None:
            step0 = step0.to(device=self.device, dtype=torch.int32).contiguous()
        if reward not in _lib.REWARD_TYPES:
            raise ValueError(f"unknown device reward {reward!r}; choose one of {[k for k in _lib.REWARD_TYPES if k]}")
        if aggregation not in _lib.AGG_MODES:
            raise ValueError(f"unknown reward aggregation {aggregation!r}; choose one of {list(_lib.AGG_MODES)}")
        if gate is not None:
            g, valid, pen, keep = self._gate(gate, B)
            _lib.check(self._lib.mpk_episode_return_gated(
                self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
                q.data_ptr(), qd.data_ptr(), C.byref(st) if st is not None else None, C.byref(g), _dptr(n_steps), seg.data_ptr(),
                _lib.REWARD_TYPES[reward], _dptr(goal), _dptr(step0), int(steps_before_reward), _lib.AGG_MODES[aggregation],
                ret.data_ptr(), B, self._stream()))
            return dict(ret=ret, seg_len=seg, done=done_out, cond_pos=cp, cond_vel=cv, valid=valid, penalty=pen)
        _lib.check(self._lib.mpk_episode_return(
            self._h, params.data_ptr(), init_pos.data_ptr(), init_vel.data_ptr(), float(init_time), C.byref(spec.c),
            q.data_ptr(), qd.data_ptr(), C.byref(st) if st is not None else None, _dptr(n_steps), seg.data_ptr(),
            _lib.REWARD_TYPES[reward], _dptr(goal), _dptr(step0), int(steps_before_reward), _lib.AGG_MODES[aggregation],
            ret.data_ptr(), B, self._stream()))
        return dict(ret=ret, seg_len=seg, done=done_out, cond_pos=cp, cond_vel=cv)

    def reward_aggregate(self, rewards: torch.Tensor, seg_len: torch.Tensor, aggregation: str = "sum") -> torch.Tensor:
        """reward_aggregation over each episode's executed steps (black_box_wrapper.py:216) of step rewards [B, T] float64, in the
        order of additions of episode_return (mpk.h: mpk_reward_aggregate): the two paths agree bit for bit"""
        B, T = rewards.shape
        assert rewards.dtype == torch.float64 and rewards.is_contiguous()
        seg_len = seg_len.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        _lib.check(self._lib.mpk_reward_aggregate(self._h, rewards.data_ptr(), seg_len.data_ptr(), _lib.AGG_MODES[aggregation],
                                                  out.data_ptr(), B, T, self._stream()))
        return out

    def pd_rollout(self, spec: RolloutSpec, des_pos: torch.Tensor, des_vel: torch.Tensor, q: torch.Tensor,
                   qd: torch.Tensor, n_steps: Optional[torch.Tensor] = None, want_actions: bool = True,
                   out: Optional[torch.Tensor] = None):
        """In-place closed-loop rollout; q, qd float64 [B, D] are updated to the state after the executed steps."""
        B, T, D = des_pos.shape
        assert des_pos.dtype == torch.float32 and des_vel.dtype == torch.float32
        assert q.dtype == torch.float64 and qd.dtype == torch.float64 and q.is_contiguous() and qd.is_contiguous()
        des_pos, des_vel = des_pos.contiguous(), des_vel.contiguous()
        act = out if out is not None else (
            torch.empty((B, T, D), dtype=torch.float32, device=self.device) if want_actions else None)
        if n_steps is not None:
            n_steps = n_steps.to(device=self.device, dtype=torch.int32).contiguous()
        if getattr(spec, "metaworld", False):
            q, qd = self._metaworld_state(q)         # static plant: the caller's state is left as it is
        _lib.check(self._lib.mpk_pd_rollout(self._h, C.byref(spec.c), des_pos.data_ptr(), des_vel.data_ptr(),
                                            q.data_ptr(), qd.data_ptr(), _dptr(n_steps), _dptr(act), B, T,
                                            self._stream()))
        return act

    def reacher_rollout(self, spec: RolloutSpec, des_pos: torch.Tensor, des_vel: torch.Tensor, q: torch.Tensor,
                        qd: torch.Tensor, goal: torch.Tensor, n_steps: Optional[torch.Tensor] = None,
                        step0: Optional[torch.Tensor] = None, steps_before_reward: int = 199,
                        want_actions: bool = True, out=None):
        """
        pd_rollout + SimpleReacherEnv's per-step reward (simple_reacher.py:56-72) on the torque double integrator:
        returns (actions float32 [B, T, D] or None, rewards float64 [B, T]); q, qd are updated in place.
        """
        self._refuse_metaworld(spec, "reacher_rollout")
        B, T, D = des_pos.shape
        assert des_pos.dtype == torch.float32 and des_vel.dtype == torch.float32
        assert q.dtype == torch.float64 and qd.dtype == torch.float64 and q.is_contiguous() and qd.is_contiguous()
        des_pos, des_vel = des_pos.contiguous(), des_vel.contiguous()
        goal = torch.as_tensor(goal, dtype=torch.float64, device=self.device).expand(B, 2).contiguous()
        if out is not None:
            act, rew = out
        else:
            act = torch.empty((B, T, D), dtype=torch.float32, device=self.device) if want_actions else None
            rew = torch.empty((B, T), dtype=torch.float64, device=self.device)
        if n_steps is not None:
            n_steps = n_steps.to(device=self.device, dtype=torch.int32).contiguous()
        if step0 is not None:
            step0 = step0.to(device=self.device, dtype=torch.int32).contiguous()
        _lib.check(self._lib.mpk_reacher_rollout(self._h, C.byref(spec.c), des_pos.data_ptr(), des_vel.data_ptr(),
                                                 q.data_ptr(), qd.data_ptr(), _dptr(n_steps), _dptr(step0),
                                                 goal.data_ptr(), int(steps_before_reward), _dptr(act),
                                                 rew.data_ptr(), B, T, self._stream()))
        return act, rew

    def condition_gather(self, pos: torch.Tensor, vel: torch.Tensor, seg_len: torch.Tensor,
                         out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """desired (pos, vel) [B, D] at the last executed step of each episode (condition_on_desired)"""
        B, T, D = pos.shape
        cp, cv = out if out is not None else (torch.empty((B, D), dtype=torch.float32, device=self.device),
                                              torch.empty((B, D), dtype=torch.float32, device=self.device))
        _lib.check(self._lib.mpk_condition_gather(self._h, pos.data_ptr(), vel.data_ptr(), seg_len.data_ptr(),
                                                  cp.data_ptr(), cv.data_ptr(), B, T, self._stream()))
        return cp, cv

    def unpin_tables(self):
        """release the basis-table slots pinned by captured graphs (mpk.h: mpk_unpin_tables); those graphs are dead then"""
        _lib.check(self._lib.mpk_unpin_tables(self._h))

    def replan_advance(self, traj_steps: torch.Tensor, plan_steps: torch.Tensor, done: torch.Tensor, every: int,
                       max_planning_times: int, horizon: int) -> torch.Tensor:
        """Integer replanning bookkeeping on device; returns seg_len int32 [B] and updates the state tensors in place."""
        B = traj_steps.shape[0]
        seg = torch.empty(B, dtype=torch.int32, device=self.device)
        mpt = int(min(max_planning_times, 2 ** 31 - 1))
        _lib.check(self._lib.mpk_replan_advance(self._h, traj_steps.data_ptr(), plan_steps.data_ptr(), seg.data_ptr(),
                                                done.data_ptr(), int(every), mpt, int(horizon), self.num_steps, B,
                                                self._stream()))
        return seg

    def traj_validity(self, pos: torch.Tensor, pos_low, pos_high, params: Optional[torch.Tensor] = None,
                      tau_bound=None, delay_bound=None, with_penalty: bool = False):
        """valid bool [B]; with_penalty: also the float64 reward of an invalid plan (table_tennis_env.py:282-289)"""
        B, T, D = pos.shape
        if with_penalty:
            lo_p, lo_k = _dvec(pos_low, D)
            hi_p, hi_k = _dvec(pos_high, D)
            check = int(params is not None and tau_bound is not None and delay_bound is not None)
            tb = (C.c_double * 2)(*(tau_bound if check else (0.0, 0.0)))
            db = (C.c_double * 2)(*(delay_bound if check else (0.0, 0.0)))
            valid = torch.empty(B, dtype=torch.uint8, device=self.device)
            pen = torch.empty(B, dtype=torch.float64, device=self.device)
            _lib.check(self._lib.mpk_traj_validity_penalty(
                self._h, pos.contiguous().data_ptr(), _dptr(params), C.cast(lo_p, C.c_void_p),
                C.cast(hi_p, C.c_void_p), check, C.cast(tb, C.c_void_p), C.cast(db, C.c_void_p), valid.data_ptr(),
                pen.data_ptr(), B, T, self._stream()))
            return valid.bool(), pen
        lo_p, lo_k = _dvec(pos_low, D)
        hi_p, hi_k = _dvec(pos_high, D)
        check = int(params is not None and tau_bound is not None and delay_bound is not None)
        tb = (C.c_double * 2)(*(tau_bound if check else (0.0, 0.0)))
        db = (C.c_double * 2)(*(delay_bound if check else (0.0, 0.0)))
        valid = torch.empty(B, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.mpk_traj_validity(self._h, pos.contiguous().data_ptr(), _dptr(params),
                                               C.cast(lo_p, C.c_void_p), C.cast(hi_p, C.c_void_p), check,
                                               C.cast(tb, C.c_void_p), C.cast(db, C.c_void_p), valid.data_ptr(), B, T,
                                               self._stream()))
        return valid.bool()

    # ---- introspection -----------------------------------------------------------------------------------------
    def prodmp_tables(self):
        n = self._lib.mpk_prodmp_tables(self._h, None, None, None, None, None, None, None)
        _lib.check(n)
        K = self.config.num_basis + 1
        arrs = [np.empty(n, np.float64) for _ in range(4)] + [np.empty((n, K), np.float64) for _ in range(2)] + \
               [np.empty(K, np.float64)]
        _lib.check(self._lib.mpk_prodmp_tables(self._h, *[a.ctypes.data for a in arrs]))
        return dict(zip(("y1", "y2", "dy1", "dy2", "pos_basis", "vel_basis", "scale"), arrs))

    def prodmp_indices(self, init_time: float = 0.0):
        T = self.num_steps
        idx = np.empty(T, np.int32)
        ib = np.empty(1, np.int32)
        _lib.check(self._lib.mpk_prodmp_indices(self._h, float(init_time), idx.ctypes.data, ib.ctypes.data,
                                                self._stream()))
        return idx, int(ib[0])

    def scaled_basis(self, times) -> np.ndarray:
        """basis functions x their parameter scale at ``times`` [n] -> float32 [n, K] (mpk.h: mpk_scaled_basis)"""
        t = np.ascontiguousarray(np.asarray(times, np.float32).reshape(-1))
        K = self.config.num_basis + (1 if self.mp_type == "prodmp" else 0)
        out = np.empty((t.shape[0], K), np.float32)
        _lib.check(self._lib.mpk_scaled_basis(self._h, t.ctypes.data, int(t.shape[0]), out.ctypes.data, self._stream()))
        return out

    def last_kernel(self) -> str:
        return self._lib.mpk_last_kernel(self._h).decode()
