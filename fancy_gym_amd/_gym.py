"""
Gymnasium compatibility layer.

The reference builds its black-box path on ``gymnasium`` (``BlackBoxWrapper`` is a ``gym.ObservationWrapper``,
``RawInterfaceWrapper`` a ``gym.Wrapper``).  When gymnasium is importable we use it unchanged; otherwise (this build
image has no gymnasium) a minimal duck-typed stand-in with the same names and semantics for the handful of features the
hot path touches is used: ``spaces.Box``, ``Env``, ``Wrapper`` (``get_wrapper_attr``, attribute forwarding, ``unwrapped``,
``spec``), ``ObservationWrapper``, ``TimeLimit``, ``register`` / ``make``.
"""
from __future__ import annotations

import copy
import importlib
from types import SimpleNamespace
from typing import Any, Callable, Dict, Optional

import numpy as np

try:  # pragma: no cover - not available in the build image
    import gymnasium as _g
    from gymnasium import spaces  # noqa: F401
    from gymnasium.wrappers import TimeLimit  # noqa: F401
    Env, Wrapper, ObservationWrapper = _g.Env, _g.Wrapper, _g.ObservationWrapper
    register, make, registry = _g.register, _g.make, _g.registry
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo_f, hi_f = np.isfinite(self.low), np.isfinite(self.high)
            out = self._rng.standard_normal(self.shape)
            both = lo_f & hi_f
            out = np.where(both, self._rng.uniform(np.where(both, self.low, 0), np.where(both, self.high, 1)), out)
            out = np.where(lo_f & ~hi_f, self.low + self._rng.exponential(size=self.shape), out)
            out = np.where(~lo_f & hi_f, self.high - self._rng.exponential(size=self.shape), out)
            return out.astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __eq__(self, other):
            return (isinstance(other, _Box) and self.shape == other.shape and np.array_equal(self.low, other.low)
                    and np.array_equal(self.high, other.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class _Dict:
        """the few features of gymnasium.spaces.Dict the path touches: ``.spaces`` (name -> space), contains, sample"""

        def __init__(self, spaces=None, seed=None, **kw):
            self.spaces = dict(spaces or {}, **kw)
            self.dtype = None
            self.shape = None

        def sample(self):
            return {k: sp.sample() for k, sp in self.spaces.items()}

        def contains(self, x):
            return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(
                self.spaces[k].contains(np.asarray(v, dtype=self.spaces[k].dtype).reshape(self.spaces[k].shape))
                for k, v in x.items())

        def __getitem__(self, k):
            return self.spaces[k]

        def __eq__(self, other):
            return isinstance(other, _Dict) and self.spaces == other.spaces

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.spaces.items()) + ")"

    spaces = SimpleNamespace(Box=_Box, Dict=_Dict)

    class Env:
        observation_space = None
        action_space = None
        spec = None

        @property
        def unwrapped(self):
            return self

        def get_wrapper_attr(self, name: str):
            return getattr(self, name)

        def reset(self, *, seed=None, options=None):
            raise NotImplementedError

        def step(self, action):
            raise NotImplementedError

        def render(self):
            return None

        def close(self):
            return None

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env
            self._action_space = None
            self._observation_space = None

        # spaces / spec forward to the wrapped env unless overridden
        @property
        def action_space(self):
            return self._action_space if self._action_space is not None else self.env.action_space

        @action_space.setter
        def action_space(self, v):
            self._action_space = v

        @property
        def observation_space(self):
            return self._observation_space if self._observation_space is not None else self.env.observation_space

        @observation_space.setter
        def observation_space(self, v):
            self._observation_space = v

        @property
        def spec(self):
            return self.env.spec

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def __getattr__(self, name):
            if name.startswith("_") or name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

        def get_wrapper_attr(self, name: str):
            if hasattr(type(self), name) or name in self.__dict__:
                return getattr(self, name)
            return self.env.get_wrapper_attr(name)

        def reset(self, *, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, action):
            return self.env.step(action)

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

    class ObservationWrapper(Wrapper):
        def reset(self, *, seed=None, options=None):
            obs, info = self.env.reset(seed=seed, options=options)
            return self.observation(obs), info

        def step(self, action):
            obs, r, term, trunc, info = self.env.step(action)
            return self.observation(obs), r, term, trunc, info

        def observation(self, observation):
            raise NotImplementedError

    class TimeLimit(Wrapper):
        def __init__(self, env, max_episode_steps: int):
            super().__init__(env)
            self._max_episode_steps = int(max_episode_steps)
            self._elapsed_steps = 0

        def reset(self, *, seed=None, options=None):
            self._elapsed_steps = 0
            return self.env.reset(seed=seed, options=options)

        def step(self, action):
            obs, r, term, trunc, info = self.env.step(action)
            self._elapsed_steps += 1
            if self._elapsed_steps >= self._max_episode_steps:
                trunc = True
            return obs, r, term, trunc, info

    class _Registry(dict):
        pass

    registry = _Registry()

    def register(id: str, entry_point=None, max_episode_steps: Optional[int] = None, kwargs: Optional[Dict] = None,
                 **_ignored):
        registry[id] = SimpleNamespace(id=id, entry_point=entry_point, max_episode_steps=max_episode_steps,
                                       kwargs=dict(kwargs or {}))

    def make(id: str, **kwargs):
        if id not in registry:
            raise ValueError(f"No registered env with id: {id}")
        spec = registry[id]
        ep = spec.entry_point
        if isinstance(ep, str):
            mod, attr = ep.split(":")
            ep = getattr(importlib.import_module(mod), attr)
        kw = copy.deepcopy(spec.kwargs)
        kw.update(kwargs)
        env = ep(**kw)
        espec = SimpleNamespace(id=id, max_episode_steps=spec.max_episode_steps, kwargs=kw)
        try:
            env.spec = espec
        except AttributeError:
            pass
        if spec.max_episode_steps:
            env = TimeLimit(env, spec.max_episode_steps)
        return env
