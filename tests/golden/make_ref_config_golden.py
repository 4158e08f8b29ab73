#!/usr/bin/env python3
"""
Reference-generated fixture for the CONFIGURATION constants of the path (build container only: reads /root/reference).

`fancy_gym` itself cannot be imported here (gymnasium / mujoco are absent), but everything the config front-end of the path does is
pure Python over literals and numpy expressions, so it is taken from the reference's files with `ast` -- nothing is imported from
the package, no stand-in modules, no reference text is stored:

  * `_BB_DEFAULTS`  = ast.literal_eval of the dict display in fancy_gym/envs/registry.py (:62-129)
  * `nested_update` = the FunctionDef of registry.py (:264-277) compiled ALONE and run (namespace: collections.abc.Mapping /
    MutableMapping, which is all it names)
  * `mp_config`     = the class-attribute dict displays of the mp_wrapper.py files, evaluated with only `np` in scope; a
    `replanning_schedule` lambda is sampled at t = 0 .. 400 into a bool vector
  * merge           = the order of bb_env_constructor (registry.py:284-292): deepcopy(defaults) <- mp_config[mp_type] <- {} <- {}
  * env facts       = MAX_EPISODE_STEPS_* constants, `frame_skip` / `repeat_action` constants (ast) and the `timestep` attribute of
    the MuJoCo XML: dt = timestep * frame_skip (* repeat_action), duration = dt * max_episode_steps (utils.get_env_duration)

Output: tests/golden/ref_configs.json -- per environment id the wrapper's own mp_config[mp_type] and the merged config (arrays as
lists, replanning schedules sampled), dt, steps, duration, plus the sha256 of every reference file read.  tests/test_oracle_pins.py compares `resolve_mp_config`, bench.py's
CFG / gains and the CFG1 - CFG5 tuples of the test-suite with it.

    python tests/golden/make_ref_config_golden.py [--check]        (--check: regenerate in memory and compare with the committed file)
"""
import ast
import copy
import hashlib
import json
import os
import re
import sys
from collections.abc import Mapping, MutableMapping

import numpy as np

REF = "/root/reference/fancy_gym"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_configs.json")

_read = {}


def src(rel):
    path = os.path.join(REF, rel)
    with open(path, "rb") as f:
        data = f.read()
    _read[rel] = hashlib.sha256(data).hexdigest()
    return data.decode()


def tree(rel):
    return ast.parse(src(rel), filename=rel)


def bb_defaults_and_nested_update():
    t = tree("envs/registry.py")
    defaults = nested = None
    for node in t.body:
        if isinstance(node, ast.Assign) and any(isinstance(x, ast.Name) and x.id == "_BB_DEFAULTS" for x in node.targets):
            defaults = ast.literal_eval(node.value)
        if isinstance(node, ast.FunctionDef) and node.name == "nested_update":
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"Mapping": Mapping, "MutableMapping": MutableMapping}
            exec(compile(mod, "registry.py:nested_update", "exec"), ns)      # the reference's own function object
            nested = ns["nested_update"]
    assert defaults is not None and nested is not None
    return defaults, nested


def class_mp_config(rel, cls):
    """the `mp_config = {...}` class attribute of `cls` in file `rel`, evaluated with numpy only"""
    for node in tree(rel).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.Assign) and any(isinstance(x, ast.Name) and x.id == "mp_config" for x in item.targets):
                    return eval(compile(ast.Expression(item.value), f"{rel}:{cls}.mp_config", "eval"), {"np": np, "__builtins__": {}})
    raise KeyError((rel, cls))


def constants(rel, name):
    """every place file `rel` binds `name` to a literal: keyword arguments, parameter defaults, dict entries, assignments -> [(value, line)]"""
    out = []
    for node in ast.walk(tree(rel)):
        if isinstance(node, ast.keyword) and node.arg == name and isinstance(node.value, ast.Constant):
            out.append((node.value.value, node.value.lineno))
        elif isinstance(node, ast.FunctionDef):
            args = node.args.args
            for a, d in zip(args[len(args) - len(node.args.defaults):], node.args.defaults):
                if a.arg == name and isinstance(d, ast.Constant):
                    out.append((d.value, d.lineno))
        elif isinstance(node, ast.Dict):
            for k, v in zip(node.keys, node.values):
                if isinstance(k, ast.Constant) and k.value == name and isinstance(v, ast.Constant):
                    out.append((v.value, v.lineno))
        elif isinstance(node, ast.Assign) and isinstance(node.value, ast.Constant):
            for tg in node.targets:
                if (isinstance(tg, ast.Name) and tg.id == name) or (isinstance(tg, ast.Attribute) and tg.attr == name):
                    out.append((node.value.value, node.lineno))
    return out


def one(rel, name):
    vals = constants(rel, name)
    assert vals and len({v for v, _ in vals}) == 1, (rel, name, vals)
    return vals[0][0], f"{rel}:{vals[0][1]}"


def xml_timestep(rel):
    m = re.search(r'timestep="([0-9.eE+-]+)"', src(rel))
    assert m, rel
    return float(m.group(1))


def jsonable(x):
    if isinstance(x, Mapping):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, np.ndarray):
        return {"__ndarray__": x.tolist(), "dtype": str(x.dtype)}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if callable(x):
        # replanning_schedule(pos, vel, obs, action, t): the schedules of the reference look at t only
        return {"__schedule__": [bool(x(None, None, None, None, t)) for t in range(401)]}
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


# id -> (mp type, mp_wrapper file, class, steps constant (file, name), xml, frame_skip (file, name) or literal, extra dt factor (file, name))
ENVS = {
    # the five BASELINE.json configurations ...
    "fancy_ProMP/Reacher5d-v0": ("ProMP", "envs/mujoco/reacher/mp_wrapper.py", "MPWrapper", ("envs/mujoco/reacher/reacher.py", "MAX_EPISODE_STEPS_REACHER"),
                                 "envs/mujoco/reacher/assets/reacher_5links.xml", ("envs/mujoco/reacher/reacher.py", "frame_skip"), None),
    "fancy_ProDMP/BoxPushingDense-v0": ("ProDMP", "envs/mujoco/box_pushing/mp_wrapper.py", "MPWrapper",
                                        ("envs/mujoco/box_pushing/box_pushing_env.py", "MAX_EPISODE_STEPS_BOX_PUSHING"),
                                        "envs/mujoco/box_pushing/assets/box_pushing.xml", ("envs/mujoco/box_pushing/box_pushing_env.py", "frame_skip"), None),
    "fancy_DMP/Reacher7d-v0": ("DMP", "envs/mujoco/reacher/mp_wrapper.py", "MPWrapper", ("envs/mujoco/reacher/reacher.py", "MAX_EPISODE_STEPS_REACHER"),
                               "envs/mujoco/reacher/assets/reacher_7links.xml", ("envs/mujoco/reacher/reacher.py", "frame_skip"), None),
    "fancy_ProDMP/BoxPushingDenseReplan-v0": ("ProDMP", "envs/mujoco/box_pushing/mp_wrapper.py", "ReplanMPWrapper",
                                              ("envs/mujoco/box_pushing/box_pushing_env.py", "MAX_EPISODE_STEPS_BOX_PUSHING"),
                                              "envs/mujoco/box_pushing/assets/box_pushing.xml",
                                              ("envs/mujoco/box_pushing/box_pushing_env.py", "frame_skip"), None),
    "fancy_ProMP/TableTennis4D-v0": ("ProMP", "envs/mujoco/table_tennis/mp_wrapper.py", "TT_MPWrapper",
                                     ("envs/mujoco/table_tennis/table_tennis_env.py", "MAX_EPISODE_STEPS_TABLE_TENNIS"),
                                     "envs/mujoco/table_tennis/assets/xml/table_tennis_env.xml", ("envs/__init__.py", "frame_skip"), None),
    # ... and the learned-phase families
    "fancy_ProDMP/TableTennis4D-v0": ("ProDMP", "envs/mujoco/table_tennis/mp_wrapper.py", "TT_MPWrapper",
                                      ("envs/mujoco/table_tennis/table_tennis_env.py", "MAX_EPISODE_STEPS_TABLE_TENNIS"),
                                      "envs/mujoco/table_tennis/assets/xml/table_tennis_env.xml", ("envs/__init__.py", "frame_skip"), None),
    # (TT_MPWrapper_Replan is what fancy/TableTennisGoalSwitchingReplan-v0 registers, envs/__init__.py:296-304; TableTennis{2,4}DReplan-v0
    # registers the plain TT_MPWrapper, :257-267)
    "fancy_ProDMP/TableTennisGoalSwitchingReplan-v0": ("ProDMP", "envs/mujoco/table_tennis/mp_wrapper.py", "TT_MPWrapper_Replan",
                                                       ("envs/mujoco/table_tennis/table_tennis_env.py", "MAX_EPISODE_STEPS_TABLE_TENNIS"),
                                                       "envs/mujoco/table_tennis/assets/xml/table_tennis_env.xml",
                                                       ("envs/mujoco/table_tennis/table_tennis_env.py", "frame_skip"), None),
    "fancy_ProMP/BeerPong-v0": ("ProMP", "envs/mujoco/beerpong/mp_wrapper.py", "MPWrapper", ("envs/mujoco/beerpong/beerpong.py", "MAX_EPISODE_STEPS_BEERPONG"),
                                "envs/mujoco/beerpong/assets/beerpong_wo_cup_big_table.xml", ("envs/mujoco/beerpong/beerpong.py", "frame_skip"),
                                ("envs/mujoco/beerpong/beerpong.py", "repeat_action")),
}


def build():
    defaults, nested_update = bb_defaults_and_nested_update()
    out = {"_BB_DEFAULTS": jsonable(defaults), "envs": {}}
    for fid, (mp_type, wfile, wcls, (sfile, sname), xml, fskip, extra) in ENVS.items():
        mp_config = class_mp_config(wfile, wcls)
        # bb_env_constructor, registry.py:284-292 (no register-time / make-time overrides for these ids)
        active = copy.deepcopy(mp_config.get(mp_type, {}))
        inherit = active.pop("inherit_defaults", mp_config.get("inherit_defaults", True))
        config = copy.deepcopy(defaults[mp_type]) if inherit else {}
        nested_update(config, active)
        nested_update(config, {})
        nested_update(config, {})
        steps, steps_at = one(sfile, sname)
        fs, fs_at = one(*fskip)
        ts = xml_timestep(xml)
        dt = ts * fs
        extra_at = None
        if extra:
            rep, extra_at = one(*extra)
            dt = dt * rep
        out["envs"][fid] = {
            "mp_type": mp_type, "mp_wrapper": f"{wfile}:{wcls}", "mp_config": jsonable(mp_config.get(mp_type, {})), "config": jsonable(config),
            "max_episode_steps": steps, "max_episode_steps_at": steps_at, "timestep": ts, "timestep_at": xml,
            "frame_skip": fs, "frame_skip_at": fs_at, "dt_factor_at": extra_at, "dt": dt, "duration": dt * steps,
        }
    # the joint limits / bounds the validity gate of TableTennis uses (table_tennis_utils.py:3-10)
    t = tree("envs/mujoco/table_tennis/table_tennis_utils.py")
    tt = {}
    for node in t.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in (
                "jnt_pos_low", "jnt_pos_high", "delay_bound", "tau_bound"):
            tt[node.targets[0].id] = jsonable(eval(compile(ast.Expression(node.value), "table_tennis_utils.py", "eval"), {"np": np, "__builtins__": {}}))
    out["table_tennis_utils"] = tt
    out["sha256"] = dict(sorted(_read.items()))
    out["provenance"] = ("generated by tests/golden/make_ref_config_golden.py from /root/reference/fancy_gym with ast (no import of the package): "
                         "literal _BB_DEFAULTS, the reference's nested_update FunctionDef compiled alone, mp_config class attributes evaluated with "
                         "numpy only, merge order of bb_env_constructor; numpy " + np.__version__)
    return out


def main():
    data = build()
    text = json.dumps(data, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        with open(OUT) as f:
            old = json.load(f)
        old.pop("provenance", None); new = json.loads(text); new.pop("provenance", None)
        assert old == new, "committed fixture differs from the reference"
        print("ref_configs.json matches the reference")
        return
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text), "bytes;", len(data["envs"]), "environment ids")


if __name__ == "__main__":
    main()
