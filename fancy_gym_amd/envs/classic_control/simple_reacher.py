"""
Host (NumPy) SimpleReacher: the one inner environment of the reference that needs no physics engine, kept here as the
single-episode counterpart of the device rollout ``TrajectoryEngine.reacher_rollout`` / ``BatchedBlackBox(reward=
"simple_reacher")`` and as the step-based env behind ``fancy/SimpleReacher-v0`` / ``fancy/LongSimpleReacher-v0``.

Behaviour follows fancy_gym/envs/classic_control (read, not copied):
  plant        torque double integrator, dt = 0.01           base_reacher/base_reacher_torque.py:20-37, base_reacher.py:21
  kinematics   planar chain of unit links, angles accumulate  base_reacher/base_reacher.py:19,97-104
  reward       -(distance to goal, paid from step 199 on) - sum(action^2)   simple_reacher/simple_reacher.py:30,56-72
  observation  [cos q, sin q, qdot, ee - goal, step]          simple_reacher/simple_reacher.py:77-85
  reset        first joint ~ U(pi/4, 3pi/4) when random_start, goal ~ U(disc of radius n_links) unless given
               (base_reacher.py:73-95, simple_reacher.py:87-99; the reference draws the goal once before and once after
               seeding, so the seeded stream gives: first-joint angle, then goal -- reproduced here)
Self-collision is only recorded by the reference (it never enters reward or termination of this env), so it is left out.
"""
from typing import Optional, Sequence

import numpy as np

from ... import _gym
from ...black_box.raw_interface_wrapper import RawInterfaceWrapper

STEPS_BEFORE_REWARD = 199
MAX_TORQUE = 1000.0


def end_effector(joint_angles: np.ndarray) -> np.ndarray:
    """tip of a chain of unit links whose joint angles are relative to the previous link; [..., D] -> [..., 2]"""
    absolute = np.cumsum(joint_angles, axis=-1)
    return np.stack([np.cumsum(np.cos(absolute), axis=-1)[..., -1], np.cumsum(np.sin(absolute), axis=-1)[..., -1]],
                    axis=-1)


class SimpleReacherEnv(_gym.Env):
    dt = 0.01

    def __init__(self, n_links: int, target: Optional[Sequence[float]] = None, random_start: bool = True,
                 render_mode: Optional[str] = None):
        self.n_links = int(n_links)
        self.fixed_target = None if target is None else np.asarray(target, dtype=np.float64)
        self.random_start = bool(random_start)
        self.render_mode = render_mode
        self.steps_before_reward = STEPS_BEFORE_REWARD
        bound = np.concatenate([np.full(2 * self.n_links, np.pi), np.full(self.n_links + 3, np.inf)])
        self.observation_space = _gym.spaces.Box(low=-bound, high=bound, shape=bound.shape)
        torque = np.full(self.n_links, MAX_TORQUE)
        self.action_space = _gym.spaces.Box(low=-torque, high=torque, shape=torque.shape)
        self.q = np.zeros(self.n_links)
        self.q[0] = np.pi / 2
        self.qd = np.zeros(self.n_links)
        self.goal = np.zeros(2)
        self.steps = 0
        self._rng = np.random.default_rng()
        self._start_pos = np.zeros(self.n_links)      # simple_reacher.py:29

    # ---- RawInterfaceWrapper plumbing ---------------------------------------------------------------------------------
    @property
    def current_pos(self) -> np.ndarray:
        return self.q.copy()

    @property
    def current_vel(self) -> np.ndarray:
        return self.qd.copy()

    # ---- episode ---------------------------------------------------------------------------------------------------------
    def _draw_goal(self) -> np.ndarray:
        if self.fixed_target is not None:
            return self.fixed_target.copy()
        reach = float(self.n_links)
        while True:
            g = self._rng.uniform(-reach, reach, size=2)
            if np.linalg.norm(g) < reach:
                return g

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        if seed is not None:
            self._rng = np.random.default_rng(seed)
        random_start = (options or {}).get("random_start", self.random_start)
        if random_start:
            self.q = np.zeros(self.n_links)
            self.q[0] = self._rng.uniform(np.pi / 4, 3 * np.pi / 4)
            self._start_pos = self.q.copy()
        else:
            # the reference starts from _start_pos: zeros in SimpleReacherEnv (simple_reacher.py:29), replaced by the
            # last random start of this instance (base_reacher.py:80-86)
            self.q = self._start_pos.copy()
        self.qd = np.zeros(self.n_links)
        self.goal = self._draw_goal()
        self.steps = 0
        return self._observe(), {}

    def _observe(self) -> np.ndarray:
        return np.concatenate([np.cos(self.q), np.sin(self.q), self.qd, end_effector(self.q) - self.goal,
                               [self.steps]]).astype(np.float32)

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        self.qd = self.qd + self.dt * action
        self.q = self.q + self.dt * self.qd
        gap = end_effector(self.q) - self.goal
        reward_dist = -float(np.sqrt(gap[0] * gap[0] + gap[1] * gap[1])) if self.steps >= self.steps_before_reward else 0.0
        reward_ctrl = float((action ** 2).sum())
        self.steps += 1
        return self._observe(), reward_dist - reward_ctrl, False, False, dict(reward_dist=reward_dist,
                                                                              reward_ctrl=reward_ctrl)


class SimpleReacherMPWrapper(RawInterfaceWrapper):
    """gains / scales of fancy_gym/envs/classic_control/simple_reacher/mp_wrapper.py:9-28, mask of :30-38"""

    mp_config = {
        "ProMP": {"controller_kwargs": {"p_gains": 0.6, "d_gains": 0.075}},
        "DMP": {"controller_kwargs": {"p_gains": 0.6, "d_gains": 0.075},
                "trajectory_generator_kwargs": {"weights_scale": 50},
                "phase_generator_kwargs": {"alpha_phase": 2}},
        "ProDMP": {},
    }

    @property
    def context_mask(self) -> np.ndarray:
        env = self.env.unwrapped
        start = [env.random_start] * (3 * env.n_links)      # cos, sin, velocity: context only when the start varies
        return np.array(start + [True, True, False])        # goal offset is context, the step counter is not

    @property
    def current_pos(self):
        return self.env.unwrapped.current_pos

    @property
    def current_vel(self):
        return self.env.unwrapped.current_vel
