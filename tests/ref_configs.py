"""
tests/golden/ref_configs.json (written by tests/golden/make_ref_config_golden.py from the reference's own files) decoded, and the
step from a merged reference config to the records the oracle / the engine take -- make_bb's rules restated
(fancy_gym/utils/make_env_helpers.py:107-126: tau defaults to the episode's duration, a learned tau / delay without bounds gets
[2 dt, duration] / [0, duration - 2 dt]); kwargs the reference does not set keep mp_pytorch's defaults (the oracle's dataclass defaults).
"""
import json
import os

import numpy as np

from oracle import mp_oracle as O

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_configs.json")
NUM_DOF = {"Reacher5d": 5, "Reacher7d": 7, "BoxPushing": 7, "TableTennis": 7, "BeerPong": 7}     # action dims


def decode(x):
    if isinstance(x, dict):
        if "__ndarray__" in x:
            return np.asarray(x["__ndarray__"], dtype=x["dtype"])
        if "__schedule__" in x:
            return np.asarray(x["__schedule__"], dtype=bool)
        return {k: decode(v) for k, v in x.items()}
    if isinstance(x, list):
        return [decode(v) for v in x]
    return x


def load():
    with open(PATH) as f:
        return decode(json.load(f))


def same(a, b) -> bool:
    """deep equality with arrays / sampled schedules"""
    if isinstance(a, dict) or isinstance(b, dict):
        return isinstance(a, dict) and isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if callable(a) or callable(b):
        fa = a if not callable(a) else np.asarray([bool(a(None, None, None, None, t)) for t in range(401)])
        fb = b if not callable(b) else np.asarray([bool(b(None, None, None, None, t)) for t in range(401)])
        return np.array_equal(fa, fb)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(np.asarray(a), np.asarray(b))
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return a == b


def num_dof(env_id: str) -> int:
    return next(d for k, d in NUM_DOF.items() if k in env_id)


def oracle_records(env_id: str, entry: dict):
    """(PhaseCfg, BasisCfg, TrajCfg, dt, duration, gains, schedule-or-None, max_planning_times-or-None) of one fixture entry"""
    cfg = entry["config"]
    dt, duration = entry["dt"], round(entry["duration"], 12)
    ph = dict(cfg["phase_generator_kwargs"]); ba = dict(cfg["basis_generator_kwargs"]); tr = dict(cfg["trajectory_generator_kwargs"])
    co = cfg["controller_kwargs"]; bb = cfg["black_box_kwargs"]
    if ph.get("tau") is None:
        ph["tau"] = duration
    if ph.get("learn_tau") and ph.get("tau_bound") is None:
        ph["tau_bound"] = [dt * 2, duration]
    if ph.get("learn_delay") and ph.get("delay_bound") is None:
        ph["delay_bound"] = [0, duration - dt * 2]
    pkw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in ph.items() if k != "phase_generator_type"}
    if not pkw.get("learn_tau"):
        pkw.pop("tau_bound", None)          # (bounds of a phase parameter that is not learned never reach the kernels)
    if not pkw.get("learn_delay"):
        pkw.pop("delay_bound", None)
    pc = O.PhaseCfg(ph["phase_generator_type"], **pkw)
    bkw = {k: v for k, v in ba.items() if k != "basis_generator_type"}
    if ba["basis_generator_type"] != "zero_rbf":
        bkw.setdefault("num_basis_zero_start", 0)
    bc = O.BasisCfg(ba["basis_generator_type"], **bkw)
    tkw = {k: v for k, v in tr.items() if k not in ("trajectory_generator_type", "duration")}
    tc = O.TrajCfg(tr["trajectory_generator_type"], action_dim=num_dof(env_id), **tkw)
    gains = (np.asarray(co["p_gains"], np.float64), np.asarray(co["d_gains"], np.float64))
    return pc, bc, tc, dt, duration, gains, bb.get("replanning_schedule"), bb.get("max_planning_times")
