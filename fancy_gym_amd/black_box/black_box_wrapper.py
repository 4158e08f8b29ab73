"""
BlackBoxWrapper -- episode-level environment: one ``step(action)`` takes a movement-primitive parameter vector, turns
it into a desired (pos, vel) trajectory on the GPU and tracks it on the wrapped step-based env.

Drop-in for ``fancy_gym/black_box/black_box_wrapper.py`` (constructor, ``step / reset / get_trajectory / render``,
public attributes).  Differences are internal only: the trajectory comes from the HIP kernels (through
``trajectory_generator``), and the code is organised as a plan / execute pair so that ``BatchedBlackBox`` can reuse the
integer bookkeeping.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np

from .._gym import ObservationWrapper, spaces
from ..utils.utils import get_numpy
from .controller.base_controller import BaseController
from .raw_interface_wrapper import RawInterfaceWrapper


def _never(*_args) -> bool:
    return False


class BlackBoxWrapper(ObservationWrapper):

    def __init__(self,
                 env: RawInterfaceWrapper,
                 trajectory_generator,
                 tracking_controller: BaseController,
                 duration: float,
                 verbose: int = 1,
                 learn_sub_trajectories: bool = False,
                 replanning_schedule: Optional[Callable[[np.ndarray, np.ndarray, np.ndarray, np.ndarray, int], bool]] = None,
                 reward_aggregation: Callable[[np.ndarray], float] = np.sum,
                 max_planning_times: int = np.inf,
                 condition_on_desired: bool = False):
        """Arguments as in the reference (black_box_wrapper.py:16-46)."""
        super().__init__(env)
        self.duration = duration
        self.learn_sub_trajectories = learn_sub_trajectories
        self.do_replanning = replanning_schedule is not None
        self.replanning_schedule = replanning_schedule or _never
        self.current_traj_steps = 0

        self.traj_gen = trajectory_generator
        self.tracking_controller = tracking_controller
        self.traj_gen.set_duration(self.duration, self.dt)

        # only learned tau / delay expose bounds (reference :60-65)
        phase = self.traj_gen.phase_gn
        self.tau_bound = getattr(phase, "tau_bound", [-np.inf, np.inf])
        self.delay_bound = getattr(phase, "delay_bound", [-np.inf, np.inf])

        self.reward_aggregation = reward_aggregation

        self.return_context_observation = not (learn_sub_trajectories or self.do_replanning)
        self.traj_gen_action_space = self._get_traj_gen_action_space()
        self.action_space = self._get_action_space()
        self.observation_space = self._get_observation_space()

        self.do_render = False
        self.verbose = verbose

        self.condition_on_desired = condition_on_desired
        self.condition_pos = None
        self.condition_vel = None

        self.max_planning_times = max_planning_times
        self.plan_steps = 0

    # ---- spaces ------------------------------------------------------------------------------------------------------
    def _get_traj_gen_action_space(self):
        low, high = self.traj_gen.get_params_bounds()
        return spaces.Box(low=get_numpy(low), high=get_numpy(high), dtype=self.env.action_space.dtype)

    def _get_action_space(self):
        """Hook for envs whose action has non-MP entries; by default the MP parameter space (reference :129-139)."""
        try:
            return self.traj_gen_action_space
        except AttributeError:
            return self._get_traj_gen_action_space()

    def _get_observation_space(self):
        if not self.return_context_observation:
            return self.env.observation_space
        mask = self.env.context_mask
        full = self.env.observation_space
        return spaces.Box(low=full.low[mask], high=full.high[mask], dtype=full.dtype)

    def observation(self, observation):
        if self.return_context_observation:
            observation = observation[self.env.context_mask]
        return observation.astype(self.observation_space.dtype)

    # ---- plan ----------------------------------------------------------------------------------------------------------
    def get_trajectory(self, action: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """MP parameters -> desired (position [T, D], velocity [T, D]); reference :96-120."""
        duration = self.duration
        if self.learn_sub_trajectories:
            duration = None
            self.traj_gen.reset()          # every sub-trajectory sets tau / delay anew
        box = self.traj_gen_action_space
        self.traj_gen.set_params(np.clip(action, box.low, box.high))
        init_time = np.array(self.current_traj_steps * self.dt if self.do_replanning else 0)
        cond_pos = self.condition_pos if self.condition_pos is not None else self.env.get_wrapper_attr("current_pos")
        cond_vel = self.condition_vel if self.condition_vel is not None else self.env.get_wrapper_attr("current_vel")
        self.traj_gen.set_initial_conditions(init_time, cond_pos, cond_vel)
        self.traj_gen.set_duration(duration, self.dt)
        return get_numpy(self.traj_gen.get_traj_pos()), get_numpy(self.traj_gen.get_traj_vel())

    # ---- execute -------------------------------------------------------------------------------------------------------
    def step(self, action: np.ndarray):
        """Generate the plan, then track it step by step on the wrapped env (reference :150-217)."""
        position, velocity = self.get_trajectory(action)
        position, velocity = self.env.set_episode_arguments(action, position, velocity)
        valid, position, velocity = self.env.preprocessing_and_validity_callback(
            action, position, velocity, self.tau_bound, self.delay_bound)

        horizon = len(position)
        rewards = np.zeros(shape=(horizon,))
        if self.verbose >= 2:
            actions = np.zeros(shape=(horizon,) + self.env.action_space.shape)
            observations = np.zeros(shape=(horizon,) + self.env.observation_space.shape,
                                    dtype=self.env.observation_space.dtype)
        infos: Dict[str, Any] = dict()
        terminated, truncated = False, False

        if not valid:
            obs, ret, terminated, truncated, infos = self.env.invalid_traj_callback(
                action, position, velocity, self.return_context_observation, self.tau_bound, self.delay_bound)
            return self.observation(obs), ret, terminated, truncated, infos

        self.plan_steps += 1
        act_space = self.env.action_space
        t, obs = -1, None
        for t, (des_pos, des_vel) in enumerate(zip(position, velocity)):
            raw = self.tracking_controller.get_action(des_pos, des_vel, self.env.get_wrapper_attr("current_pos"),
                                                      self.env.get_wrapper_attr("current_vel"))
            c_action = np.clip(raw, act_space.low, act_space.high)
            obs, c_reward, terminated, truncated, info = self.env.step(c_action)
            rewards[t] = c_reward
            if self.verbose >= 2:
                actions[t, :] = c_action
                observations[t, :] = obs
            for k, v in info.items():
                infos.setdefault(k, [None] * horizon)[t] = v
            if self.do_render:
                self.env.render()
            replan = (not (terminated or truncated)) and self.replanning_schedule(
                self.env.get_wrapper_attr("current_pos"), self.env.get_wrapper_attr("current_vel"), obs, c_action,
                t + 1 + self.current_traj_steps) and self.plan_steps < self.max_planning_times
            if terminated or truncated or replan:
                if self.condition_on_desired:
                    self.condition_pos, self.condition_vel = des_pos, des_vel
                break

        executed = t + 1
        infos.update({k: v[:executed] for k, v in infos.items()})
        self.current_traj_steps += executed
        if self.verbose >= 2:
            infos["positions"] = position
            infos["velocities"] = velocity
            infos["step_actions"] = actions[:executed]
            infos["step_observations"] = observations[:executed]
            infos["step_rewards"] = rewards[:executed]
        infos["trajectory_length"] = executed
        return self.observation(obs), self.reward_aggregation(rewards[:executed]), terminated, truncated, infos

    def render(self):
        self.do_render = True

    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        self.current_traj_steps = 0
        self.plan_steps = 0
        self.traj_gen.reset()
        self.condition_pos = None
        self.condition_vel = None
        return super().reset(seed=seed, options=options)
