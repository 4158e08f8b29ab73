"""TimeAwareObservation (reference utils/wrappers.py:11-87): appends t / max_steps to the observation; added
automatically for replanning / sub-trajectory envs (make_env_helpers.py:95-97)."""
import copy

import numpy as np

from .._gym import ObservationWrapper, spaces


class TimeAwareObservation(ObservationWrapper):
    """
    The observation space of the wrapped environment is a flat ``Box`` -- the progress ``t / max_episode_steps`` is
    appended as one more entry with bounds [0, 1] (utils/wrappers.py:33-39,57-58) -- or a ``Dict`` -- it becomes the entry
    ``'time_awareness'``, a float64 ``Box(0, 1)`` (:40-47,59-62).  ``enforce_dtype_float32`` asserts a float32 observation
    space (:27-30).
    """

    def __init__(self, env, enforce_dtype_float32: bool = False):
        super().__init__(env)
        space = env.observation_space
        if enforce_dtype_float32:
            assert space.dtype == np.float32, \
                "TimeAwareObservation was given an environment with a dtype!=np.float32 (" + str(space.dtype) + \
                "). This requirement can be removed by setting enforce_dtype_float32=False."
        self._is_dict = isinstance(space, spaces.Dict)
        assert self._is_dict or isinstance(space, spaces.Box), str(space) + " is not supported. Only Box or Dict"
        if self._is_dict:
            sub = copy.copy(space.spaces)
            sub["time_awareness"] = spaces.Box(0, 1, dtype=np.float64)
            self.observation_space = spaces.Dict(sub)
        else:
            low = np.append(space.low, 0.0)
            high = np.append(space.high, 1.0)
            self.observation_space = spaces.Box(low, high, dtype=space.dtype)
        self.t = 0
        spec = getattr(env, "spec", None)
        self._max_episode_steps = getattr(spec, "max_episode_steps", None) or 1
        self.is_vector_env = getattr(env, "is_vector_env", False)

    def observation(self, observation):
        if self._is_dict:
            obs = copy.copy(observation)
            obs["time_awareness"] = self.t / self._max_episode_steps
            return obs
        return np.append(observation, self.t / self._max_episode_steps)

    def step(self, action):
        self.t += 1
        return super().step(action)

    def reset(self, *, seed=None, options=None):
        self.t = 0
        return super().reset(seed=seed, options=options)
