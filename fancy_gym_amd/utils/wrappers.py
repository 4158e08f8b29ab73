"""TimeAwareObservation (reference utils/wrappers.py:11-87): appends t / max_steps to the observation; added
automatically for replanning / sub-trajectory envs (make_env_helpers.py:95-97)."""
import numpy as np

from .._gym import ObservationWrapper, spaces


class TimeAwareObservation(ObservationWrapper):
    def __init__(self, env):
        super().__init__(env)
        box = env.observation_space
        low = np.append(box.low, 0.0)
        high = np.append(box.high, 1.0)
        self.observation_space = spaces.Box(low, high, dtype=box.dtype)
        self.t = 0
        spec = getattr(env, "spec", None)
        self._max_episode_steps = getattr(spec, "max_episode_steps", None) or 1

    def observation(self, observation):
        return np.append(observation, self.t / self._max_episode_steps)

    def step(self, action):
        self.t += 1
        return super().step(action)

    def reset(self, *, seed=None, options=None):
        self.t = 0
        return super().reset(seed=seed, options=options)
