"""
Config front-end of the black-box path: ``_BB_DEFAULTS`` per MP type, the ``mp_config`` merge rules and the
``<ns>_<MP>/<name>`` id scheme (reference fancy_gym/envs/registry.py:62-129,137-309).  Only what the hot path needs:
no environment zoo is registered here -- users ``register`` / ``upgrade`` their own step-based envs.
"""
from __future__ import annotations

import copy
import importlib
from collections.abc import Mapping, MutableMapping
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np

from .. import _gym
from ..black_box.raw_interface_wrapper import RawInterfaceWrapper
from ..utils.make_env_helpers import make_bb


class DefaultMPWrapper(RawInterfaceWrapper):
    """uses env.context_mask / env.current_pos / env.current_vel when the env defines them (reference :18-59)"""

    @property
    def context_mask(self):
        if hasattr(self.env, "context_mask"):
            return self.env.context_mask
        return np.full(self.env.observation_space.shape, True)

    @property
    def current_pos(self) -> Union[float, int, np.ndarray, Tuple]:
        assert hasattr(self.env, "current_pos"), \
            "DefaultMPWrapper was unable to access env.current_pos. Please write a custom MPWrapper (recommended) or expose this attribute directly."
        return self.env.current_pos

    @property
    def current_vel(self) -> Union[float, int, np.ndarray, Tuple]:
        assert hasattr(self.env, "current_vel"), \
            "DefaultMPWrapper was unable to access env.current_vel. Please write a custom MPWrapper (recommended) or expose this attribute directly."
        return self.env.current_vel


def _motor(p=1.0, d=0.1):
    return {"controller_type": "motor", "p_gains": p, "d_gains": d}


# values: reference registry.py:62-129
_BB_DEFAULTS = {
    "ProMP": {
        "wrappers": [],
        "trajectory_generator_kwargs": {"trajectory_generator_type": "promp"},
        "phase_generator_kwargs": {"phase_generator_type": "linear"},
        "controller_kwargs": _motor(),
        "basis_generator_kwargs": {"basis_generator_type": "zero_rbf", "num_basis": 5, "num_basis_zero_start": 1,
                                   "basis_bandwidth_factor": 3.0},
        "black_box_kwargs": {},
    },
    "DMP": {
        "wrappers": [],
        "trajectory_generator_kwargs": {"trajectory_generator_type": "dmp"},
        "phase_generator_kwargs": {"phase_generator_type": "exp"},
        "controller_kwargs": _motor(),
        "basis_generator_kwargs": {"basis_generator_type": "rbf", "num_basis": 5},
        "black_box_kwargs": {},
    },
    "ProDMP": {
        "wrappers": [],
        "trajectory_generator_kwargs": {"trajectory_generator_type": "prodmp", "duration": 2.0, "weights_scale": 1.0},
        "phase_generator_kwargs": {"phase_generator_type": "exp", "tau": 1.5},
        "controller_kwargs": _motor(),
        "basis_generator_kwargs": {"basis_generator_type": "prodmp", "alpha": 10, "num_basis": 5},
        "black_box_kwargs": {},
    },
}

KNOWN_MPS = list(_BB_DEFAULTS.keys())
_KNOWN_MPS_PLUS_ALL = KNOWN_MPS + ["all"]
ALL_MOVEMENT_PRIMITIVE_ENVIRONMENTS = {mp_type: [] for mp_type in _KNOWN_MPS_PLUS_ALL}
MOVEMENT_PRIMITIVE_ENVIRONMENTS_FOR_NS: Dict[str, Dict[str, List[str]]] = {}


def nested_update(base: MutableMapping, update):
    """
    Recursive dict update with the reference's quirk (registry.py:264-277): a sub-dict of the update that names a
    ``*_type`` REPLACES the corresponding base sub-dict wholesale (choosing another generator type drops the defaults
    of the previous one).
    """
    if any(item.endswith("_type") for item in update):
        return update
    for k, v in update.items():
        base[k] = nested_update(base.get(k, {}), v) if isinstance(v, Mapping) else v
    return base


def resolve_mp_config(mp_type: str, mp_config: Optional[Mapping] = None, register_override: Optional[Mapping] = None,
                      make_override: Optional[Mapping] = None) -> Dict[str, Any]:
    """defaults <- wrapper's mp_config[mp_type] <- register-time override <- make-time override (registry.py:284-292)"""
    mp_config = mp_config or {}
    active = copy.deepcopy(mp_config.get(mp_type, {}))
    inherit = active.pop("inherit_defaults", mp_config.get("inherit_defaults", True))
    config = copy.deepcopy(_BB_DEFAULTS[mp_type]) if inherit else {}
    nested_update(config, active)
    nested_update(config, register_override or {})
    nested_update(config, make_override or {})
    return config


def bb_env_constructor(underlying_id, mp_wrapper, mp_type, mp_config_override={}, _mp_config_override_register={},
                       **kwargs):
    raw_env = _gym.make(underlying_id, **kwargs)
    env = mp_wrapper(raw_env)
    config = resolve_mp_config(mp_type, getattr(env, "mp_config", {}), _mp_config_override_register,
                               mp_config_override)
    wrappers = config.pop("wrappers")
    return make_bb(env,
                   wrappers=wrappers,
                   black_box_kwargs=config.pop("black_box_kwargs", {}),
                   traj_gen_kwargs=config.pop("trajectory_generator_kwargs", {}),
                   controller_kwargs=config.pop("controller_kwargs", {}),
                   phase_kwargs=config.pop("phase_generator_kwargs", {}),
                   basis_kwargs=config.pop("basis_generator_kwargs", {}),
                   **config)


def fancy_id(id: str, mp_type: str) -> Tuple[str, str, str]:
    parts = id.split("/")
    if len(parts) == 1:
        ns, name = "gym", parts[0]
    elif len(parts) == 2:
        ns, name = parts
    else:
        raise ValueError('env id can not contain multiple "/".')
    tail = name.split("-")
    assert len(tail) >= 2 and tail[-1].startswith("v"), "Malformed env id, must end in -v{int}."
    return ns, name, f"{ns}_{mp_type}/{name}"


def register_mp(id: str, base_id: str, mp_wrapper, mp_type: str, mp_config_override: Dict[str, Any] = {}):
    assert mp_type in KNOWN_MPS, "Unknown mp_type"
    assert id not in ALL_MOVEMENT_PRIMITIVE_ENVIRONMENTS[mp_type], \
        f"The environment {id} is already registered for {mp_type}."
    ns, _name, fid = fancy_id(id, mp_type)
    _gym.register(id=fid, entry_point=bb_env_constructor,
                  kwargs={"underlying_id": base_id, "mp_wrapper": mp_wrapper, "mp_type": mp_type,
                          "_mp_config_override_register": mp_config_override})
    ALL_MOVEMENT_PRIMITIVE_ENVIRONMENTS[mp_type].append(fid)
    ALL_MOVEMENT_PRIMITIVE_ENVIRONMENTS["all"].append(fid)
    per_ns = MOVEMENT_PRIMITIVE_ENVIRONMENTS_FOR_NS.setdefault(ns, {m: [] for m in _KNOWN_MPS_PLUS_ALL})
    per_ns[mp_type].append(fid)
    per_ns["all"].append(fid)


def upgrade(id: str, mp_wrapper=DefaultMPWrapper, add_mp_types: List[str] = KNOWN_MPS, base_id: Optional[str] = None,
            mp_config_override: Dict[str, Any] = {}):
    """register the MP versions of an already registered step-based env (reference :186-220)"""
    for mp_type in add_mp_types:
        register_mp(id, base_id or id, mp_wrapper, mp_type, mp_config_override.get(mp_type, {}))


def register(id: str, entry_point: Optional[Union[Callable, str]] = None, mp_wrapper=DefaultMPWrapper,
             register_step_based: bool = True, add_mp_types: List[str] = KNOWN_MPS,
             mp_config_override: Dict[str, Any] = {}, **kwargs):
    """register a step-based env together with its MP versions (reference :137-183)"""
    if register_step_based:
        assert entry_point is not None, "You need to provide an entry-point, when registering step-based."
    if not callable(mp_wrapper):
        mod_name, attr_name = mp_wrapper.split(":")
        mp_wrapper = getattr(importlib.import_module(mod_name), attr_name)
    if register_step_based:
        _gym.register(id=id, entry_point=entry_point, **kwargs)
    upgrade(id, mp_wrapper, add_mp_types, mp_config_override=mp_config_override)
