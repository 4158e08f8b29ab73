"""
Gymnasium-free clones of the reference's fixture envs (test/test_black_box.py:27-56): ToyEnv (1-D Box obs/action,
dt = 0.02, reward == 1, never terminates; registered as 'toy-v0' with max_episode_steps = 50) and ToyWrapper
(current_pos == 1, current_vel == 0); plus the reference's torque double integrator
(fancy_gym/envs/classic_control/base_reacher/base_reacher_torque.py:20-37) as a D-DoF host env.
"""
import numpy as np

from fancy_gym_amd import _gym
from fancy_gym_amd.black_box.raw_interface_wrapper import RawInterfaceWrapper


class ToyEnv(_gym.Env):
    dt = 0.02

    def __init__(self, a: int = 0, b: float = 0.0, c: list = [], d: dict = {}, dim: int = 1):
        self.a, self.b, self.c, self.d = a, b, c, d
        self.observation_space = _gym.spaces.Box(low=-1, high=1, shape=(1,), dtype=np.float64)
        self.action_space = _gym.spaces.Box(low=-1, high=1, shape=(dim,), dtype=np.float64)

    def reset(self, *, seed=None, options=None):
        return np.array([-1.0]), {}

    def step(self, action):
        return np.array([-1.0]), 1, False, False, {"toy": 7}


class ToyWrapper(RawInterfaceWrapper):
    @property
    def current_pos(self):
        return np.ones(self.action_space.shape)

    @property
    def current_vel(self):
        return np.zeros(self.action_space.shape)


class DoubleIntegratorEnv(_gym.Env):
    """vel += dt * a; pos += dt * vel (base_reacher_torque.py:25-26), D DoF, reward = -|pos|^2"""
    dt = 0.02

    def __init__(self, dim: int = 7, max_torque: float = 1.0):
        self.dim = dim
        self.observation_space = _gym.spaces.Box(low=-np.inf, high=np.inf, shape=(2 * dim,), dtype=np.float64)
        self.action_space = _gym.spaces.Box(low=-max_torque, high=max_torque, shape=(dim,), dtype=np.float64)
        self.pos = np.zeros(dim)
        self.vel = np.zeros(dim)

    def reset(self, *, seed=None, options=None):
        rng = np.random.default_rng(seed)
        self.pos = rng.uniform(-1, 1, self.dim)
        self.vel = np.zeros(self.dim)
        return np.concatenate([self.pos, self.vel]), {}

    def step(self, action):
        self.vel = self.vel + self.dt * action
        self.pos = self.pos + self.dt * self.vel
        return np.concatenate([self.pos, self.vel]), -float(np.sum(self.pos ** 2)), False, False, {}


class DoubleIntegratorWrapper(RawInterfaceWrapper):
    @property
    def current_pos(self):
        return self.env.unwrapped.pos.copy()

    @property
    def current_vel(self):
        return self.env.unwrapped.vel.copy()


def register_toys():
    if "toy-v0" not in _gym.registry:
        _gym.register(id="toy-v0", entry_point="tests.toy_env:ToyEnv", max_episode_steps=50)
    if "dint-v0" not in _gym.registry:
        _gym.register(id="dint-v0", entry_point="tests.toy_env:DoubleIntegratorEnv", max_episode_steps=100)
