"""
make_bb: wire a step-based env, its RawInterfaceWrapper and the four generator / controller objects into a
BlackBoxWrapper (reference utils/make_env_helpers.py:68-136; same signature and derived defaults).
"""
from __future__ import annotations

from collections.abc import MutableMapping
from typing import Iterable, Type, Union

import numpy as np

from .. import _gym
from ..black_box.black_box_wrapper import BlackBoxWrapper
from ..black_box.factory import (get_basis_generator, get_controller, get_phase_generator,
                                 get_trajectory_generator)
from ..black_box.raw_interface_wrapper import RawInterfaceWrapper
from .wrappers import TimeAwareObservation


def ensure_finite_time(env, fallback_max_steps=500):
    spec = getattr(env, "spec", None)
    if not getattr(spec, "max_episode_steps", None):
        limit = getattr(env.unwrapped, "max_path_length", None) or fallback_max_steps
        wrapped = _gym.TimeLimit(env, limit)
        if not _gym.HAVE_GYMNASIUM and spec is not None:
            spec.max_episode_steps = limit
        return wrapped
    return env


def _make_wrapped_env(env, wrappers: Iterable[Type], fallback_max_steps=None):
    """apply `wrappers`; at least one RawInterfaceWrapper must end up in the chain (reference :35-65)"""
    if fallback_max_steps:
        env = ensure_finite_time(env, fallback_max_steps)
    found = False
    head = env
    while hasattr(head, "env"):
        if isinstance(head, RawInterfaceWrapper):
            found = True
            break
        head = head.env
    for w in wrappers:
        found = found or issubclass(w, RawInterfaceWrapper)
        env = w(env)
    if not found:
        raise ValueError("A RawInterfaceWrapper is required in order to leverage movement primitive environments.")
    return env


def get_env_duration(env) -> float:
    return env.spec.max_episode_steps * env.dt


def _verify_time_limit(mp_time_limit, env_time_limit):
    if mp_time_limit is not None and env_time_limit is not None:
        assert mp_time_limit == env_time_limit, \
            f"The specified 'time_limit' of {env_time_limit}s does not match the duration of {mp_time_limit}s for the MP."


def make_bb(env: Union[object, str], wrappers: Iterable, black_box_kwargs: MutableMapping,
            traj_gen_kwargs: MutableMapping, controller_kwargs: MutableMapping, phase_kwargs: MutableMapping,
            basis_kwargs: MutableMapping, time_limit: int = None, fallback_max_steps: int = None, **kwargs):
    _verify_time_limit(traj_gen_kwargs.get("duration"), time_limit)

    sub_trajs = black_box_kwargs.get("learn_sub_trajectories")
    replanning = black_box_kwargs.get("replanning_schedule")
    if sub_trajs and replanning:
        raise ValueError("Cannot used sub-trajectory learning and replanning together.")

    wrappers = list(wrappers)
    if (sub_trajs or replanning) and not any(issubclass(w, TimeAwareObservation) for w in wrappers):
        wrappers.insert(0, TimeAwareObservation)   # first, so that it alters the observation

    if isinstance(env, str):
        env = _gym.make(env, **kwargs)
    env = _make_wrapped_env(env=env, wrappers=wrappers, fallback_max_steps=fallback_max_steps)

    traj_gen_kwargs["action_dim"] = traj_gen_kwargs.get("action_dim", int(np.prod(env.action_space.shape)))
    if black_box_kwargs.get("duration") is None:
        black_box_kwargs["duration"] = get_env_duration(env)
    if phase_kwargs.get("tau") is None:
        phase_kwargs["tau"] = black_box_kwargs["duration"]
    if sub_trajs is not None:
        phase_kwargs["learn_tau"] = True       # sub-trajectories must learn their length (reference :115-117)
    # at least two env steps, otherwise the finite-difference velocity does not exist (reference :119-126)
    if phase_kwargs.get("learn_tau") and phase_kwargs.get("tau_bound") is None:
        phase_kwargs["tau_bound"] = [env.dt * 2, black_box_kwargs["duration"]]
    if phase_kwargs.get("learn_delay") and phase_kwargs.get("delay_bound") is None:
        phase_kwargs["delay_bound"] = [0, black_box_kwargs["duration"] - env.dt * 2]

    phase_gen = get_phase_generator(**phase_kwargs)
    basis_gen = get_basis_generator(phase_generator=phase_gen, **basis_kwargs)
    controller = get_controller(**controller_kwargs)
    traj_gen = get_trajectory_generator(basis_generator=basis_gen, **traj_gen_kwargs)
    return BlackBoxWrapper(env, trajectory_generator=traj_gen, tracking_controller=controller, **black_box_kwargs)
