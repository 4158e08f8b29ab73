"""
BatchedBlackBox -- the batched sibling of BlackBoxWrapper: B independent episodes of ONE movement-primitive
configuration evaluated at once on one GPU (the reference always runs B = 1: black_box_wrapper.py:96-120,175-203).

One ``step(params[B, P])`` does what ``BlackBoxWrapper.step`` does for every episode:
    plan      get_trajectory: clip -> (frozen tau/delay) -> boundary conditions -> (pos, vel)[B, T, D]     [HIP]
    validity  optional joint-limit / bound check (raw_interface_wrapper.py:55-72)                         [HIP]
    schedule  integer replanning bookkeeping for the schedule ``t % every == 0``                            [HIP]
    execute   tracking controller + plant loop for the executed steps (plants that live on the GPU)        [HIP]
and returns everything as CUDA tensors.  Environments that cannot live on the GPU (MuJoCo) consume ``des_pos`` /
``des_vel`` on the host instead (``plant=None``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from .black_box.controller.base_controller import BaseController
from .engine import RolloutSpec, TrajectoryEngine
from .mp.traj import MPInterface


class BatchedBlackBox:

    def __init__(self, trajectory_generator: MPInterface, tracking_controller: BaseController, num_envs: int,
                 dt: float, duration: float, act_low=-math.inf, act_high=math.inf,
                 plant: Optional[str] = "double_integrator", replanning_every: Optional[int] = None,
                 max_planning_times: Union[int, float] = math.inf, condition_on_desired: bool = False,
                 max_episode_steps: Optional[int] = None, pos_limits: Optional[Sequence] = None,
                 check_tau_delay: bool = False, reward: Optional[str] = None, steps_before_reward: int = 199,
                 device=None, learn_sub_trajectories: bool = False, reward_aggregation="sum", verbose: int = 2):
        """
        trajectory_generator / tracking_controller: the objects the factories return (``get_trajectory_generator``,
        ``get_controller``).  ``replanning_every = n`` is the schedule ``lambda pos, vel, obs, action, t: t % n == 0``
        (e.g. envs/mujoco/box_pushing/mp_wrapper.py:89).  ``pos_limits = (low[D], high[D])`` enables the batched
        validity check (envs/mujoco/table_tennis/table_tennis_env.py:303-309).  ``reward = "simple_reacher"`` adds the
        per-step reward of the reference's SimpleReacher (envs/classic_control/simple_reacher/simple_reacher.py:56-72)
        to the device rollout: ``step`` then also returns ``step_rewards [B, T]`` and their sum ``rewards [B]``
        (``reward_aggregation``: "sum" / "mean" / "last" or np.sum / np.mean -- black_box_wrapper.py:24,216 -- over the EXECUTED
        steps of each episode, on the device); goals are given to ``reset``.

        ``learn_sub_trajectories`` (black_box_wrapper.py:98-102, utils/make_env_helpers.py:89-117): every ``step`` plans a new
        sub-trajectory of ``round(tau / dt)`` steps from the current state -- tau is the first parameter (``learn_tau``), read
        anew by every plan (nothing is frozen), clipped to its bounds; an episode ends when ``max_episode_steps`` are done.  On
        the device a plan is a row mask over the max-length plan: episode b executes the first ``round(tau_b / dt)`` rows
        (``trajectory_length``), the rows behind them are not part of its plan.  (A single-episode wrapper evaluates the plan on
        the grid ``linspace(0, T_b dt, T_b + 1)[1:]``, the batch on the first T_b points of the grid of the longest plan: the
        same times up to one fp32 rounding of the grid, exactly the same whenever the grid values are exact in fp32.)
        Not with replanning (make_env_helpers.py:91-92).

        Arbitrary ``replanning_schedule`` callables see host state per step and stay with the single-episode wrapper; the
        device schedule is ``t % replanning_every == 0``.

        ``verbose`` (black_box_wrapper.py:21,160,184,208-213): 2 (the default HERE: this class has always returned everything)
        ``step`` returns the plans, step actions and step rewards [B, T, .] like ``infos`` of a verbose = 2 wrapper; < 2 (the
        reference's default is 1) it returns what ``BlackBoxWrapper.step`` returns then -- the aggregated reward, the flags,
        ``trajectory_length`` and the plant state -- and with a device plant the whole step is ONE launch that stores nothing per
        step (mpk_episode_return): plan, controller, plant, reward and aggregation on the CU.  Same state bit for bit; the same
        aggregated rewards bit for bit wherever the verbose = 2 path writes its step rewards with the tile kernel (every shape this
        path accepts; the per-episode fallback kernel behind "pd_generic" 1 / D = 1 sums a step's squared actions as a tree: 1e-13
        relative -- mpk.h, mpk_episode_return).  Falls back to the verbose = 2 launches (and drops their arrays)
        where the fused kernel does not apply: sub-trajectories, a device reward together with a learned phase, drifted episodes.
        """
        self.verbose = int(verbose)
        self._lean_ok = True
        self.traj_gen = trajectory_generator
        self.tracking_controller = tracking_controller
        self.B, self.dt, self.duration = int(num_envs), float(dt), float(duration)
        if device is not None:
            self.traj_gen._device = device
        self.traj_gen.set_duration(self.duration, self.dt)
        self.engine: TrajectoryEngine = self.traj_gen.engine()
        self.device = self.engine.device
        self.D, self.T = self.engine.num_dof, self.engine.num_steps
        self.horizon = int(max_episode_steps) if max_episode_steps is not None else self.T
        self.learn_sub_trajectories = bool(learn_sub_trajectories)
        if self.learn_sub_trajectories and replanning_every is not None:
            raise ValueError("Cannot used sub-trajectory learning and replanning together.")      # make_env_helpers.py:91-92
        if self.learn_sub_trajectories and not self.traj_gen.phase_gn.learn_tau:
            raise ValueError("learn_sub_trajectories needs a learned tau (make_bb sets learn_tau: make_env_helpers.py:110-112)")
        agg = {np.sum: "sum", np.mean: "mean"}.get(reward_aggregation, reward_aggregation)
        if agg not in ("sum", "mean", "last"):
            raise ValueError(f"reward_aggregation must be 'sum', 'mean', 'last', np.sum or np.mean on the device, got {reward_aggregation!r}")
        self.reward_aggregation = agg
        self.do_replanning = replanning_every is not None
        self.every = int(replanning_every) if self.do_replanning else self.horizon + 1
        self.max_planning_times = max_planning_times
        self.condition_on_desired = bool(condition_on_desired)
        self.plant = plant
        self.pos_limits = pos_limits
        self.check_tau_delay = bool(check_tau_delay)
        if reward not in (None, "simple_reacher"):
            raise ValueError(f"unknown device reward {reward!r}")
        if reward is not None and plant != "double_integrator":
            raise ValueError("the simple_reacher reward needs plant='double_integrator'")
        self.reward = reward
        self.steps_before_reward = int(steps_before_reward)
        self.goal = None
        ctype = getattr(tracking_controller, "device_type", None)
        self.spec = None
        if plant is not None:
            if ctype is None:
                raise ValueError(f"{type(tracking_controller).__name__} has no device implementation; use plant=None "
                                 f"and step the environments on the host")
            self.spec = RolloutSpec(ctype, self.D, getattr(tracking_controller, "p_gains", 0.0),
                                    getattr(tracking_controller, "d_gains", 0.0), act_low, act_high, plant=plant,
                                    dt=self.dt)
        phase = self.traj_gen.phase_gn
        self.tau_bound = getattr(phase, "tau_bound", [-np.inf, np.inf])
        self.delay_bound = getattr(phase, "delay_bound", [-np.inf, np.inf])
        self._n_phase = int(phase.learn_tau) + int(phase.learn_delay)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.traj_steps = torch.zeros(self.B, **i32)
        self.plan_steps = torch.zeros(self.B, **i32)
        self.done = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self._prev_done, self._prev_done_known = None, True
        self.q = torch.zeros((self.B, self.D), dtype=torch.float64, device=self.device)
        self.qd = torch.zeros_like(self.q)
        self.condition_pos = None
        self.condition_vel = None
        self._frozen_phase = None
        # traj_steps of every live episode while the schedule keeps them in lockstep; None = per-episode init_time from the
        # device counters.  With the validity gate an invalid plan takes its episode out of lockstep, which the host can only
        # learn by reading the device back; `device_time` (set by capture_episode for a gated episode) runs the gate on
        # per-episode times from the first plan on instead: nothing synchronises, the step is a fixed sequence of launches
        self.device_time = False
        self._lockstep = 0
        self._host_plans = 0
        self._const_flags = None    # (all-True, all-False) [B], shared by every fused step's result
        self._phase_bounds = None   # [2, n_phase] bounds of the learned tau / delay, on the device
        self._plans_since_reset = 0
        self._start32 = None        # fp32 image of the plant state at reset (boundary condition of the first plan)

    # ---- episode control ---------------------------------------------------------------------------------------------
    def check_range(self):
        """
        ProDMP with a per-episode phase (learned tau / delay, drifted init_time): a plan whose scaled time leaves the
        pre-computed table range raises RuntimeError in mp_pytorch and in the single-episode BlackBoxWrapper; the batched
        kernels can only raise a device flag (and clamp the index).  This synchronises and raises that RuntimeError if any
        plan since the last check left the range.  ``reset`` calls it for the episodes just finished (skipped while a
        hipGraph is being captured -- call it yourself after a replay).
        """
        self.engine.check_range()

    def _range_can_overflow(self) -> bool:
        cfg = self.engine.config
        return self.engine.mp_type == "prodmp" and bool(cfg.learn_tau or cfg.learn_delay or self._lockstep is None)

    def reset(self, init_pos=None, init_vel=None, goal=None):
        """start B new episodes from plant state (init_pos, init_vel) [B, D] (default zeros); goal [B, 2] for the
        simple_reacher reward"""
        if self._plans_since_reset and self._range_can_overflow() and not torch.cuda.is_current_stream_capturing():
            self.check_range()
        elif self._plans_since_reset:
            # the episodes just finished: whoever read their results has synchronised; a ring kernel that gave up waiting in their LAST
            # plan would otherwise be reported by the next launch only (costs nothing: the fault word lives in host memory)
            self.engine.poll_fault()
        if self.reward is not None:
            if goal is None:
                raise ValueError("reward='simple_reacher' needs goal [B, 2] at reset")
            self.goal = torch.as_tensor(goal, dtype=torch.float64, device=self.device).expand(self.B, 2).contiguous()
        # one launch (mpk_episode_reset): integer state, plant state and its fp32 image (the first plan's boundary state)
        def state(x):
            if x is None:
                return None
            x = torch.as_tensor(x, dtype=torch.float64, device=self.device)
            return x.contiguous() if tuple(x.shape) == (self.B, self.D) else x.expand(self.B, self.D).contiguous()
        if self._start32 is None:
            self._start32 = tuple(torch.empty((self.B, self.D), dtype=torch.float32, device=self.device)
                                  for _ in range(2))
        self.engine.episode_reset(self.q, self.qd, self.traj_steps, self.plan_steps, self.done, state(init_pos),
                                  state(init_vel), cond=self._start32)
        self.condition_pos = self.condition_vel = None
        self._frozen_phase = None
        self._lockstep = None if (self.device_time and self.do_replanning) else 0
        self._host_plans = 0
        self._plans_since_reset = 0
        self._prev_done = None              # the done bytes before the next plan: none is done (mpk_gate_flags takes NULL)
        self._prev_done_known = True
        self.traj_gen.reset()
        return self.q, self.qd

    @property
    def current_pos(self) -> torch.Tensor:
        return self.q

    @property
    def current_vel(self) -> torch.Tensor:
        return self.qd

    def params_bounds(self) -> np.ndarray:
        return self.engine.params_bounds()

    # ---- plan ----------------------------------------------------------------------------------------------------------
    def _plan_params(self, params) -> torch.Tensor:
        """[B, P] float32 on the device, with the phase parameters the episode froze at its first plan"""
        params = torch.as_tensor(params, dtype=torch.float32, device=self.device)
        if params.shape != (self.B, self.engine.num_params):
            raise ValueError(f"params must be [{self.B}, {self.engine.num_params}], got {tuple(params.shape)}")
        if self._n_phase and self.learn_sub_trajectories:
            # every sub-trajectory sets tau / delay anew (the reference resets the generator: black_box_wrapper.py:99-102);
            # the kernels clip them to their bounds
            return params.contiguous()
        if self._n_phase:
            # tau / delay are frozen by the first plan of an episode (mp_pytorch 'finalize'; pinned by
            # test/test_replanning_sequencing.py:231-335): later plans reuse them
            if self._frozen_phase is None:
                if self._phase_bounds is None:
                    self._phase_bounds = torch.as_tensor(self.engine.params_bounds()[:, :self._n_phase],
                                                         device=self.device)
                lo = self._phase_bounds
                self._frozen_phase = torch.minimum(torch.maximum(params[:, :self._n_phase], lo[0]), lo[1]).to(torch.float32)
            # (one launch -- a clone and a slice assignment were two; the plan's own columns follow the frozen ones)
            params = torch.cat((self._frozen_phase, params[:, self._n_phase:]), dim=1)
        return params

    def get_trajectory(self, params) -> Dict[str, torch.Tensor]:
        params = self._plan_params(params)
        cond_pos = self.condition_pos if self.condition_pos is not None else self.q.float()
        cond_vel = self.condition_vel if self.condition_vel is not None else self.qd.float()
        if self.do_replanning and self._lockstep is None:
            init_time = (self.traj_steps.double() * self.dt).float()     # per-episode
        else:
            init_time = float(self._lockstep * self.dt) if self.do_replanning else 0.0
        pos, vel = self.engine.trajectory(params, cond_pos, cond_vel, init_time)
        if self.learn_sub_trajectories and self.engine.mp_type == "promp":
            # ProMP's velocity is the forward difference of its positions with the LAST row repeating the one before
            # (mp_pytorch; make_env_helpers.py:119-122): the last row of a sub-trajectory of T_b steps is row T_b - 2, not the
            # difference towards a step T_b the sub-trajectory does not have
            n = self._plan_length(params).to(torch.int64)
            rows = torch.arange(self.B, device=self.device)
            vel[rows, n - 1] = vel[rows, (n - 2).clamp(min=0)]
        return {"params": params, "des_pos": pos, "des_vel": vel}

    def _plan_length(self, params: torch.Tensor) -> torch.Tensor:
        """learn_sub_trajectories: round(clip(tau_b) / dt) steps (np.round and torch.round: half to even), int32 [B]"""
        tau = params[:, 0].clamp(float(self.tau_bound[0]), float(self.tau_bound[1]))     # what the kernels use (fp32 clip)
        return torch.round(tau.double() / self.dt).to(torch.int32).clamp(1, self.T)

    # ---- plan + execute ----------------------------------------------------------------------------------------------
    def _host_segment(self) -> int:
        """host mirror of the integer rule for episodes that move in lockstep (no device read-back)"""
        cur = self._lockstep
        if cur >= self.horizon:
            return 0
        self._host_plans += 1
        g_break = self.horizon
        if self._host_plans < self.max_planning_times:
            g_break = min((cur // self.every + 1) * self.every, self.horizon)
        return max(1, min(g_break - cur, self.T))

    def _can_fuse(self) -> bool:
        """plan + execute through mpk_replan_step(_gated) (one launch for promp / prodmp with a shared OR a learned phase and for
        dmp on its response route, the separate kernels otherwise): needs the device plant, no device reward, and episodes that
        still move in lockstep (one init_time for all).  Round 6: the validity gate runs inside the same launch."""
        return (self.spec is not None and self.plant == "double_integrator"
                and self.reward is None and not self.learn_sub_trajectories
                and (not self.do_replanning or self._lockstep is not None))

    def _gate(self, raw_params):
        """the validity gate of this wrapper as the engine takes it (mpk.h: mpk_validity_gate); None without pos_limits"""
        if self.pos_limits is None:
            return None
        return dict(pos_low=self.pos_limits[0], pos_high=self.pos_limits[1], check_tau_delay=self.check_tau_delay,
                    tau_bound=self.tau_bound, delay_bound=self.delay_bound,
                    raw_params=torch.as_tensor(raw_params, dtype=torch.float32, device=self.device) if self.check_tau_delay else None)

    def _was_done(self) -> Optional[torch.Tensor]:
        """the done bytes before this plan WITHOUT a launch: the snapshot the step before returned (the one-launch steps write it), None
        right after a reset (none is done); a copy only after a step of the separate-launch path"""
        if getattr(self, "_prev_done_known", False):
            return self._prev_done
        return self.done.clone()

    def _step_fused(self, params) -> Dict[str, torch.Tensor]:
        """plan + execute as ONE device operation (mpk_replan_step: integer state, trajectory + rollout, condition gather
        in a single launch where the fused closed-loop kernel applies)"""
        gate = self._gate(params)           # (the RAW action: the reference checks tau / delay before clipping, table_tennis_env.py:305-306)
        was_done = self._was_done() if gate is not None else None
        params = self._plan_params(params)
        first = self._start32 is not None and self._plans_since_reset == 1    # q, qd untouched since reset
        cond_pos = self.condition_pos if self.condition_pos is not None else (self._start32[0] if first else self.q.float())
        cond_vel = self.condition_vel if self.condition_vel is not None else (self._start32[1] if first else self.qd.float())
        init_time = float(self._lockstep * self.dt) if self.do_replanning else 0.0
        mpt = self.max_planning_times if math.isfinite(self.max_planning_times) else 2 ** 31 - 1
        r = self.engine.replan_step(params, cond_pos, cond_vel, self.spec, self.q, self.qd, self.traj_steps,
                                    self.plan_steps, self.done, self.every, int(mpt), self.horizon,
                                    init_time=init_time, condition=self.condition_on_desired, gate=gate)
        seg = r["seg_len"]
        if self.condition_on_desired:
            self.condition_pos, self.condition_vel = r["cond_pos"], r["cond_vel"]
        if self.do_replanning:
            # the host mirrors the integer rule -- with the gate too: an invalid plan FINISHES its episode, so every episode
            # that is still live has executed exactly the segments the rule gives (nothing is read back from the device)
            self._lockstep += self._host_segment()
        done = r["done"].view(torch.bool)               # 0 / 1 bytes: a view, not a launch
        self._prev_done, self._prev_done_known = r["done"], True
        if gate is not None:
            valid = r["valid"].view(torch.bool)
            # invalid plans terminate their episode without executing a step (black_box_wrapper.py:169-172); both flags in one launch
            terminated, truncated = self.engine.gate_flags(r["valid"], was_done, r["done"])
            return dict(params=params, des_pos=r["pos"], des_vel=r["vel"], step_actions=r["actions"], valid=valid,
                        invalid_penalty=r["penalty"], trajectory_length=seg, done=done, terminated=terminated,
                        truncated=truncated, current_pos=self.q, current_vel=self.qd)
        if self._const_flags is None:
            self._const_flags = (torch.ones(self.B, dtype=torch.bool, device=self.device),
                                 torch.zeros(self.B, dtype=torch.bool, device=self.device))
        valid, never = self._const_flags
        # nothing can invalidate a plan on this path: terminated stays False, truncated is `done`
        return dict(params=params, des_pos=r["pos"], des_vel=r["vel"], step_actions=r["actions"], valid=valid,
                    trajectory_length=seg, done=done, terminated=never, truncated=done, current_pos=self.q,
                    current_vel=self.qd)

    def _can_episode_return(self) -> bool:
        return (self.verbose < 2 and self._lean_ok and self.spec is not None and self.plant == "double_integrator"
                and not self.learn_sub_trajectories and (self._n_phase == 0 or self.reward is None)
                and (not self.do_replanning or self._lockstep is not None))

    def _step_lean(self, params) -> Optional[Dict[str, torch.Tensor]]:
        """the verbose < 2 step as ONE launch without per-step outputs (mpk_episode_return); None = not available here"""
        gate = self._gate(params)
        was_done = self._was_done() if gate is not None else None
        params = self._plan_params(params)
        first = self._start32 is not None and self._plans_since_reset == 1
        cond_pos = self.condition_pos if self.condition_pos is not None else (self._start32[0] if first else self.q.float())
        cond_vel = self.condition_vel if self.condition_vel is not None else (self._start32[1] if first else self.qd.float())
        init_time = float(self._lockstep * self.dt) if self.do_replanning else 0.0
        mpt = self.max_planning_times if math.isfinite(self.max_planning_times) else 2 ** 31 - 1
        try:
            r = self.engine.episode_return(params, cond_pos, cond_vel, self.spec, self.q, self.qd,
                                           replan=(self.traj_steps, self.plan_steps, self.done, self.every, int(mpt), self.horizon),
                                           reward=self.reward, goal=self.goal, steps_before_reward=self.steps_before_reward,
                                           aggregation=self.reward_aggregation, init_time=init_time,
                                           condition=self.condition_on_desired, gate=gate)
        except NotImplementedError:
            self._lean_ok = False
            return None
        if self.condition_on_desired:
            self.condition_pos, self.condition_vel = r["cond_pos"], r["cond_vel"]
        if self.do_replanning:
            self._lockstep += self._host_segment()
        done = r["done"].view(torch.bool)
        self._prev_done, self._prev_done_known = r["done"], True
        if self._const_flags is None:
            self._const_flags = (torch.ones(self.B, dtype=torch.bool, device=self.device),
                                 torch.zeros(self.B, dtype=torch.bool, device=self.device))
        valid, never = self._const_flags
        out = dict(params=params, valid=valid, trajectory_length=r["seg_len"], done=done, terminated=never, truncated=done,
                   current_pos=self.q, current_vel=self.qd)
        if gate is not None:
            valid = r["valid"].view(torch.bool)
            terminated, truncated = self.engine.gate_flags(r["valid"], was_done, r["done"])
            out.update(valid=valid, invalid_penalty=r["penalty"], terminated=terminated, truncated=truncated)
        if self.reward is not None:
            out["rewards"] = r["ret"]
        return out

    def _finish(self, out, seg, valid, was_done) -> Dict[str, torch.Tensor]:
        pos, vel = out["des_pos"], out["des_vel"]
        out.update(valid=valid, trajectory_length=seg, done=self.done.bool(), terminated=~valid & ~was_done,
                   truncated=self.done.bool() & valid)
        if self.condition_on_desired and not self.learn_sub_trajectories:
            # (black_box_wrapper.py:197-203 stores the desired state only inside the break branch -- terminated, truncated, or the
            # replanning schedule.  With sub-trajectories the schedule is never true: a plan that ends before the episode does leaves
            # condition_pos None, and the next one starts from env.current_pos / current_vel -- self.q / self.qd here.)
            self.condition_pos, self.condition_vel = self.engine.condition_gather(pos, vel, seg)
        if self.do_replanning and self._lockstep is not None:
            if self.pos_limits is None:
                # no validity gate: every episode follows the same integer sequence, which the host can mirror without
                # reading the device state back (k_replan_advance's rule, black_box_wrapper.py:174,197,206)
                self._lockstep += self._host_segment()
            else:
                live = ~self.done.bool() | ~was_done
                s = seg[live]
                if s.numel() and bool((s == s[0]).all()) and bool(valid.all()):
                    self._lockstep += int(s[0])
                else:
                    self._lockstep = None      # episodes drifted apart: per-episode init_time from now on
        out.update(current_pos=self.q, current_vel=self.qd)
        return out

    _PER_STEP = ("des_pos", "des_vel", "step_actions", "step_rewards")

    def step(self, params, fuse: bool = True) -> Dict[str, torch.Tensor]:
        self._plans_since_reset += 1
        if fuse and self._can_episode_return():
            out = self._step_lean(params)
            if out is not None:
                return out
        out = self._step_full(params, fuse)
        if self.verbose < 2:        # (the launches of the verbose = 2 path ran: what it keeps per step is simply not returned)
            for k in self._PER_STEP:
                out.pop(k, None)
        return out

    def _step_full(self, params, fuse: bool = True) -> Dict[str, torch.Tensor]:
        if fuse and self._can_fuse():
            return self._step_fused(params)
        out = self.get_trajectory(params)
        pos, vel = out["des_pos"], out["des_vel"]
        was_done = self.done.bool()
        self._prev_done_known = False       # (this path changes the done bytes with launches of its own: _was_done copies them next time)
        valid = torch.ones(self.B, dtype=torch.bool, device=self.device)
        if self.pos_limits is not None:
            # `params` as the caller passed them: the reference checks the raw action, not the clipped one
            raw = torch.as_tensor(params, dtype=torch.float32, device=self.device)
            valid, out["invalid_penalty"] = self.engine.traj_validity(
                pos, self.pos_limits[0], self.pos_limits[1], raw if self.check_tau_delay else None,
                self.tau_bound if self.check_tau_delay else None,
                self.delay_bound if self.check_tau_delay else None, with_penalty=True)
            # invalid plans terminate their episode without executing a step (black_box_wrapper.py:169-172)
            self.done |= (~valid).to(torch.uint8)
        mpt = self.max_planning_times if math.isfinite(self.max_planning_times) else 2 ** 31 - 1
        if self.learn_sub_trajectories:
            seg = self._sub_trajectory_advance(out["params"])
        else:
            seg = self.engine.replan_advance(self.traj_steps, self.plan_steps, self.done, self.every, int(mpt),
                                             self.horizon)
        if self.reward is not None:
            act, rew = self.engine.reacher_rollout(self.spec, pos, vel, self.q, self.qd, self.goal, n_steps=seg,
                                                   step0=self.traj_steps - seg,
                                                   steps_before_reward=self.steps_before_reward)
            out.update(step_actions=act, step_rewards=rew, rewards=self._aggregate(rew, seg))
        elif self.spec is not None:
            out["step_actions"] = self.engine.pd_rollout(self.spec, pos, vel, self.q, self.qd, n_steps=seg)
        return self._finish(out, seg, valid, was_done)


    def _sub_trajectory_advance(self, params: torch.Tensor) -> torch.Tensor:
        """
        learn_sub_trajectories: the integer part of one step, on the device (a handful of elementwise launches, nothing read
        back).  Episode b plans round(clip(tau_b) / dt) steps (np.round: half to even, as torch.round) and executes them unless
        its step budget ends first (the TimeLimit of the reference's step-based env truncates: test/test_replanning_sequencing.
        py:100-109); finished episodes are left alone.  Returns the executed steps int32 [B].
        """
        plan_len = self._plan_length(params)
        live = self.done == 0
        left = (self.horizon - self.traj_steps).clamp(min=0)
        seg = torch.where(live, torch.minimum(plan_len, left), torch.zeros_like(plan_len))
        self.traj_steps += seg
        self.plan_steps += live.to(torch.int32)
        self.done |= (self.traj_steps >= self.horizon).to(torch.uint8)
        return seg

    def _aggregate(self, rew: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
        """reward_aggregation(rewards[:t + 1]) of black_box_wrapper.py:216 for every episode: sum / mean / last over its
        executed steps (step_rewards are zero behind them); an episode that executed nothing gets 0"""
        # one launch, in the order of additions of the verbose < 2 kernel (mpk_reward_aggregate): the two paths agree bit for bit
        return self.engine.reward_aggregate(rew, seg, self.reward_aggregation)

    # ---- whole episodes as one hipGraph ----------------------------------------------------------------------------------
    def capture_episode(self, n_plans: int, with_goal: bool = False) -> "EpisodeGraph":
        """
        Capture ``reset`` + ``n_plans`` calls of ``step`` into one hipGraph.  At B of a few thousand a plan costs ~100 us of
        Python / ctypes / allocator work around ~20 us of kernels; a replay pays one graph launch for the whole
        episode.  Requirement: a device-resident plant (``plant != None``).  With the validity gate the episodes run on
        per-episode times from the device counters (``device_time``: an invalid plan takes its episode out of lockstep, which
        only the device knows), so that nothing synchronises during capture either.

        Write the inputs into the returned object's static buffers (``init_pos``, ``init_vel``, ``params[k]``, ``goal``),
        call ``replay()``, read ``outs[k]`` (the dicts ``step`` returned during capture; their tensors are rewritten by
        every replay).
        """
        if self.spec is None:
            raise ValueError("capture_episode needs a device plant (host environments cannot be captured)")
        if self.pos_limits is not None and not (self.plant == "double_integrator" and self.reward is None
                                                and not self.learn_sub_trajectories):
            self.device_time = True         # (the fused, gated step keeps the host's lockstep mirror: nothing to read back)
        return EpisodeGraph(self, int(n_plans), with_goal)


class EpisodeGraph:
    def __init__(self, bb: BatchedBlackBox, n_plans: int, with_goal: bool):
        self.bb = bb
        dev = bb.device
        self.init_pos = torch.zeros((bb.B, bb.D), dtype=torch.float64, device=dev)
        self.init_vel = torch.zeros((bb.B, bb.D), dtype=torch.float64, device=dev)
        self.goal = torch.zeros((bb.B, 2), dtype=torch.float64, device=dev) if (with_goal or bb.reward) else None
        self.params = [torch.zeros((bb.B, bb.engine.num_params), dtype=torch.float32, device=dev)
                       for _ in range(n_plans)]
        self.outs = []

        def episode():
            kw = {"goal": self.goal} if self.goal is not None else {}
            bb.reset(self.init_pos, self.init_vel, **kw)
            return [bb.step(p) for p in self.params]

        # one eager pass on a side stream first (allocator warm-up, lazy initialisation), then the capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            episode()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outs = episode()
        torch.cuda.synchronize(dev)

    def replay(self):
        self.graph.replay()
        return self.outs
