#!/usr/bin/env python3
"""
ONE-COMMAND ORACLE PINNING.      python tools/pin_against_mp_pytorch.py [--out tests/golden] [--tol 1e-5] [--check-only]

The arithmetic of the path lives in the un-vendored third-party package ``mp_pytorch`` (pinned ``<=0.1.3`` by the
reference: pyproject.toml:30); it cannot be installed in the build container, so oracle/mp_oracle.py is a restatement
whose parity is UNPINNED.  This script is what closes that gap for a maintainer who HAS the package:

  1. builds the five BASELINE configurations + the reference's TableTennis ProDMP configuration (learned tau / delay) +
     one small probe per "(?)" switch of SURVEY Appendix A with **mp_pytorch itself**, through exactly the calls the
     reference makes: the three factories (fancy_gym/black_box/factory/*.py: ``LinearPhaseGenerator`` /
     ``ExpDecayPhaseGenerator``, ``NormalizedRBFBasisGenerator`` / ``ZeroPaddingNormalizedRBFBasisGenerator`` /
     ``ProDMPBasisGenerator``, ``ProMP`` / ``DMP`` / ``ProDMP``) and the call sequence of
     ``BlackBoxWrapper.get_trajectory`` (black_box_wrapper.py:96-120): ``set_duration`` (at construction, :57) ->
     clip to ``get_params_bounds()`` -> ``set_params`` -> ``set_initial_conditions(init_time, pos, vel)`` ->
     ``set_duration(duration, dt)`` -> ``get_traj_pos()`` / ``get_traj_vel()``, one episode per call (B = 1);
  2. writes ``<out>/ref_<case>.npz``: the inputs, mp_pytorch's (pos, vel), the package / torch / numpy versions and the
     sha256 of this script -- REFERENCE OUTPUTS, the first ones this repository would hold;
  3. evaluates oracle/mp_oracle.py on the same inputs under every setting of the switches that matter for the case
     (``relative_goal_mode``, ``goal_offset_mode``, ``single_rbf_mode``, ``dmp_first_sample``) and prints, per case, the
     error of each setting and which one matches at ``--tol`` (relative to max|reference| per array);
  4. exits 0 iff every case has a matching setting and all cases agree on ONE setting per switch; prints that setting
     (what to put into ``mpk_config`` / the host classes; include/mpk.h:60-75).

``--check-only`` skips (1)-(2) and re-evaluates (3)-(4) against ``ref_*.npz`` files that are already there: that is what
``tests/test_oracle_pins.py::test_oracle_against_mp_pytorch_reference_outputs`` runs on every machine once the fixtures
are committed.  Runs in the BUILD container / on a maintainer's machine only: it imports the reference's dependency and
never travels to the GPU box as anything but the data files it wrote.

``--package NAME`` imports the generators from another top-level package with the same layout
(``NAME.phase_gn``, ``NAME.basis_gn``, ``NAME.mp``).  tests/ uses that with ``tests.fake_mp_pytorch`` -- an mp_pytorch-
shaped facade over the oracle -- to test THIS SCRIPT's plumbing (call sequence, switch detection); outputs produced that
way are marked ``package = tests.fake_mp_pytorch`` and pin nothing.
"""
from __future__ import annotations

import argparse
import dataclasses
import hashlib
import importlib
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import mp_oracle as O  # noqa: E402
from tests.golden import make_golden as G  # noqa: E402  (the BASELINE configuration records and the seeded inputs)

# the four "(?)" switches (oracle dataclass field, owner record, candidate values; first = the shipped default)
SWITCHES = {
    "relative_goal_mode": ("tc", ("before_scale", "after_scale")),
    "goal_offset_mode": ("tc", ("ignore", "add")),
    "single_rbf_mode": ("bc", ("unit_gap", "refuse")),
    "dmp_first_sample": ("tc", ("init", "step")),
}


def _cases():
    """name -> dict(pc, bc, tc, dt, duration, B, init_times, extra_traj_kwargs, switches)"""
    cases = {}
    for name, cfg in G.CONFIGS.items():
        c = dict(cfg)
        c["extra"] = {}
        c["switches"] = []
        t = cfg["tc"].trajectory_generator_type
        if t == "dmp":
            c["switches"] = ["dmp_first_sample"]
        if t == "prodmp":
            c["switches"] = ["relative_goal_mode"] if cfg["tc"].relative_goal else []
        cases[name] = c
    # the reference passes goal_offset = 1.0 for these two (box_pushing/mp_wrapper.py:77, table_tennis/mp_wrapper.py:114)
    for name in ("cfg4_prodmp_replan", "tt_prodmp_learn_tau_delay"):
        cases[name]["extra"] = {"goal_offset": 1.0}
        cases[name]["switches"] = cases[name]["switches"] + ["goal_offset_mode"]
    # ---- one probe per switch: small, and built so that the two settings differ by far more than the tolerance ------------
    cases["probe_relative_goal"] = dict(
        pc=O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0), bc=O.BasisCfg("prodmp", num_basis=4, basis_bandwidth_factor=2, alpha=10),
        tc=O.TrajCfg("prodmp", action_dim=3, weights_scale=0.7, goal_scale=0.5, relative_goal=True),
        dt=0.02, duration=1.5, B=2, init_times=[0.0, 0.4], extra={}, switches=["relative_goal_mode"])
    cases["probe_goal_offset"] = dict(
        pc=O.PhaseCfg("exp", tau=1.5, alpha_phase=3.0), bc=O.BasisCfg("prodmp", num_basis=4, basis_bandwidth_factor=2, alpha=10),
        tc=O.TrajCfg("prodmp", action_dim=3), dt=0.02, duration=1.5, B=2, init_times=[0.0],
        extra={"goal_offset": 1.0}, switches=["goal_offset_mode"])
    cases["probe_single_rbf"] = dict(
        pc=O.PhaseCfg("linear", tau=2.0), bc=O.BasisCfg("rbf", num_basis=1, basis_bandwidth_factor=3),
        tc=O.TrajCfg("promp", action_dim=2), dt=0.02, duration=2.0, B=2, init_times=[0.0], extra={},
        switches=["single_rbf_mode"])
    # ---- ProMP corners nothing but recollection covers (no switch behind them: either the oracle's RBF placement matches or
    # it does not): centres OUTSIDE [delay, delay + tau] (num_basis_outside > 0: first centre at -outside * gap, the last gap
    # repeated for the bandwidth), and zero padding at the GOAL end (num_basis_zero_goal > 0: the returned columns are the
    # middle ones, normalised over all nb + zs + zg).  Both with a non-zero delay so that the time -> phase mapping of the
    # centres is exercised as well.
    cases["probe_promp_basis_outside"] = dict(
        pc=O.PhaseCfg("linear", tau=1.6, delay=0.2), bc=O.BasisCfg("rbf", num_basis=6, basis_bandwidth_factor=3, num_basis_outside=2),
        tc=O.TrajCfg("promp", action_dim=2, weights_scale=0.8), dt=0.02, duration=2.0, B=2, init_times=[0.0], extra={},
        switches=[])
    cases["probe_promp_zero_goal"] = dict(
        pc=O.PhaseCfg("linear", tau=1.6, delay=0.2),
        bc=O.BasisCfg("zero_rbf", num_basis=4, basis_bandwidth_factor=3, num_basis_zero_start=2, num_basis_zero_goal=3),
        tc=O.TrajCfg("promp", action_dim=2), dt=0.02, duration=2.0, B=2, init_times=[0.0], extra={}, switches=[])
    cases["probe_dmp_first_sample"] = dict(
        pc=O.PhaseCfg("exp", tau=2.0, alpha_phase=2.0), bc=O.BasisCfg("rbf", num_basis=4, basis_bandwidth_factor=3),
        tc=O.TrajCfg("dmp", action_dim=2, alpha=25.0), dt=0.02, duration=2.0, B=2, init_times=[0.0, 0.5], extra={},
        switches=["dmp_first_sample"])
    return cases


# ----------------------------------------------------------------------------------------------------------------------
# (1) the reference package, driven the way the reference drives it
# ----------------------------------------------------------------------------------------------------------------------
def build_generators(pkg: str, case):
    """fancy_gym/utils/make_env_helpers.py:128-131 -> the three factories, with the kwargs the registry merge produces"""
    phase_gn = importlib.import_module(pkg + ".phase_gn")
    basis_gn = importlib.import_module(pkg + ".basis_gn")
    mp = importlib.import_module(pkg + ".mp")
    pc, bc, tc = case["pc"], case["bc"], case["tc"]
    # phase (factory/phase_generator_factory.py:9-23)
    pk = dict(tau=pc.tau, delay=pc.delay, learn_tau=pc.learn_tau, learn_delay=pc.learn_delay)
    if pc.learn_tau or np.isfinite(pc.tau_bound[1]) or pc.tau_bound[0] != 1e-5:
        pk["tau_bound"] = list(pc.tau_bound)
    if pc.learn_delay or np.isfinite(pc.delay_bound[1]) or pc.delay_bound[0] != 0.0:
        pk["delay_bound"] = list(pc.delay_bound)
    if pc.phase_generator_type == "exp":
        phase = phase_gn.ExpDecayPhaseGenerator(alpha_phase=pc.alpha_phase, **pk)
    else:
        phase = phase_gn.LinearPhaseGenerator(**pk)
    # basis (factory/basis_generator_factory.py:10-17)
    if bc.basis_generator_type == "rbf":
        basis = basis_gn.NormalizedRBFBasisGenerator(phase, num_basis=bc.num_basis,
                                                     basis_bandwidth_factor=bc.basis_bandwidth_factor,
                                                     num_basis_outside=bc.num_basis_outside)
    elif bc.basis_generator_type == "zero_rbf":
        basis = basis_gn.ZeroPaddingNormalizedRBFBasisGenerator(phase, num_basis=bc.num_basis,
                                                                num_basis_zero_start=bc.num_basis_zero_start,
                                                                num_basis_zero_goal=bc.num_basis_zero_goal,
                                                                basis_bandwidth_factor=bc.basis_bandwidth_factor)
    else:
        basis = basis_gn.ProDMPBasisGenerator(phase, num_basis=bc.num_basis,
                                              basis_bandwidth_factor=bc.basis_bandwidth_factor, alpha=bc.alpha)
    # trajectory generator (factory/trajectory_generator_factory.py:11-18): positional (basis, action_dim), rest kwargs
    t = tc.trajectory_generator_type
    if t == "promp":
        traj = mp.ProMP(basis, tc.action_dim, weights_scale=tc.weights_scale, **case["extra"])
    elif t == "dmp":
        traj = mp.DMP(basis, tc.action_dim, weights_scale=tc.weights_scale, goal_scale=tc.goal_scale, alpha=tc.alpha,
                      **case["extra"])
    else:
        traj = mp.ProDMP(basis, tc.action_dim, weights_scale=tc.weights_scale, goal_scale=tc.goal_scale,
                         auto_scale_basis=tc.auto_scale_basis, relative_goal=tc.relative_goal,
                         disable_goal=tc.disable_goal, disable_weights=tc.disable_weights, **case["extra"])
    return traj


def _numpy(x):
    """fancy_gym/utils/utils.py:27-36 get_numpy"""
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def reference_outputs(pkg: str, case, params, ip, iv):
    """BlackBoxWrapper.get_trajectory (black_box_wrapper.py:96-120), one episode per call; returns {k: (pos, vel)} or the
    exception the package raised (a switch may be 'the package refuses this')"""
    out = {}
    try:
        traj = build_generators(pkg, case)
        traj.set_duration(case["duration"], case["dt"])                      # black_box_wrapper.py:57
        low, high = (_numpy(b) for b in traj.get_params_bounds())            # :124-125
        for k, it in enumerate(case["init_times"]):
            pos, vel = [], []
            for b in range(params.shape[0]):
                clipped = np.clip(params[b], low, high)                      # :104-105
                traj.set_params(clipped)                                     # :106
                traj.set_initial_conditions(np.array(it), ip[b], iv[b])      # :107-114
                traj.set_duration(case["duration"], case["dt"])              # :115
                pos.append(np.array(_numpy(traj.get_traj_pos()), np.float32))   # :117
                vel.append(np.array(_numpy(traj.get_traj_vel()), np.float32))   # :118
            out[k] = (np.stack(pos), np.stack(vel))
    except Exception as e:  # noqa: BLE001 - recorded, the comparison decides what it means
        return {"raised": f"{type(e).__name__}: {e}"}
    return out


# ----------------------------------------------------------------------------------------------------------------------
# (3) the oracle under every setting of the switches that matter
# ----------------------------------------------------------------------------------------------------------------------
def oracle_outputs(case, setting, params, ip, iv):
    pc = case["pc"]
    bc = dataclasses.replace(case["bc"], **{k: v for k, v in setting.items() if SWITCHES[k][0] == "bc"})
    tc = dataclasses.replace(case["tc"], **{k: v for k, v in setting.items() if SWITCHES[k][0] == "tc"})
    if "goal_offset" in case["extra"]:
        tc = dataclasses.replace(tc, goal_offset=float(case["extra"]["goal_offset"]))
    out = {}
    try:
        for k, it in enumerate(case["init_times"]):
            out[k] = O.get_trajectory(pc, bc, tc, params, case["duration"], case["dt"], it, ip, iv, dtype=np.float32)
    except Exception as e:  # noqa: BLE001
        return {"raised": f"{type(e).__name__}: {e}"}
    return out


def rel_err(ref, got, dt):
    """
    max over the init times of max|got - ref| / max|ref|, for (pos, vel); inf when exactly one side raised.  Third value:
    what fp32 itself allows the velocity to differ by -- velocities are (ProMP: literally) differences of fp32 positions
    over dt, so two correct fp32 evaluations in different operation orders differ by up to 2 ulp(max|pos|) / dt
    (the same allowance tests/test_gpu_trajectory.py grants the GPU path).
    """
    if ("raised" in ref) != ("raised" in got):
        return float("inf"), float("inf"), 0.0
    if "raised" in ref:
        return 0.0, 0.0, 0.0
    ep = ev = allow = 0.0
    for k in ref:
        (rp, rv), (gp, gv) = ref[k], got[k]
        if rp.shape != gp.shape:
            return float("inf"), float("inf"), 0.0
        sp, sv = max(np.abs(rp).max(), 1e-30), max(np.abs(rv).max(), 1e-30)
        ep = max(ep, float(np.abs(gp - rp).max() / sp))
        ev = max(ev, float(np.abs(gv - rv).max() / sv))
        allow = max(allow, float(2 * np.spacing(np.float32(sp)) / dt / sv))
    return ep, ev, allow


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--tol", type=float, default=1e-5,
                    help="relative tolerance that decides MATCH (default: north_star's 1e-5 contract; the settings of a "
                         "switch differ by > 1e-2 on the probes).  Errors are printed raw, and matches that also hold at "
                         "1e-6 are marked so")
    ap.add_argument("--package", default="mp_pytorch")
    ap.add_argument("--check-only", action="store_true", help="diff the oracle against ref_*.npz that already exist")
    args = ap.parse_args(argv)
    cases = _cases()
    refs = {}
    if not args.check_only:
        try:
            pkg = importlib.import_module(args.package)
        except ImportError as e:
            print(f"[pin] cannot import {args.package}: {e}\n[pin] install the reference's pin (`pip install 'mp_pytorch<=0.1.3'`) "
                  f"and re-run; nothing was written, parity stays UNPINNED", file=sys.stderr)
            return 2
        import torch
        version = getattr(pkg, "__version__", None)
        if version is None:
            try:
                from importlib.metadata import version as _v
                version = _v(args.package.split(".")[0])
            except Exception:  # noqa: BLE001
                version = "unknown"
        os.makedirs(args.out, exist_ok=True)
        for name, case in cases.items():
            params, ip, iv = G.make_inputs(case["pc"], case["bc"], case["tc"], case["B"], seed=len(name))
            ref = reference_outputs(args.package, case, params, ip, iv)
            save = dict(params=params, init_pos=ip, init_vel=iv, init_times=np.array(case["init_times"], np.float64),
                        package=np.array(args.package), package_version=np.array(str(version)),
                        versions=np.array(f"numpy {np.__version__}; torch {torch.__version__}; python {sys.version.split()[0]}"),
                        generator_sha256=np.array(_sha(os.path.abspath(__file__))),
                        meta=np.array(f"outputs of {args.package} {version} driven through the reference's factory + "
                                      "BlackBoxWrapper.get_trajectory call sequence (tools/pin_against_mp_pytorch.py)"))
            if "raised" in ref:
                save["raised"] = np.array(ref["raised"])
            else:
                for k, (p, v) in ref.items():
                    save[f"pos_{k}"], save[f"vel_{k}"] = p, v
            np.savez_compressed(os.path.join(args.out, f"ref_{name}.npz"), **save)
            refs[name] = (ref, params, ip, iv, f"{args.package} {version}")
    else:
        for name in cases:
            path = os.path.join(args.out, f"ref_{name}.npz")
            if not os.path.exists(path):
                continue
            z = np.load(path)
            if "raised" in z.files:
                ref = {"raised": str(z["raised"])}
            else:
                ref = {k: (z[f"pos_{k}"], z[f"vel_{k}"]) for k in range(len(z["init_times"]))}
            refs[name] = (ref, z["params"], z["init_pos"], z["init_vel"], f"{z['package']} {z['package_version']}")
        if not refs:
            print(f"[pin] no ref_*.npz under {args.out}: run this script once where mp_pytorch is importable", file=sys.stderr)
            return 2

    # ---- (3) + (4) -----------------------------------------------------------------------------------------------------
    print(f"{'case':30s} {'switch setting':46s} {'pos err':>9s} {'vel err':>9s}  verdict   (tol {args.tol:g}, reference = "
          f"{next(iter(refs.values()))[4]})")
    votes = {k: set(v[1]) for k, v in SWITCHES.items()}      # settings still compatible with every case
    all_ok = True
    for name, (ref, params, ip, iv, _) in refs.items():
        case = cases[name]
        names = case["switches"]
        combos = [dict(zip(names, vals)) for vals in itertools.product(*(SWITCHES[s][1] for s in names))] or [{}]
        matching = []
        for setting in combos:
            ep, ev, allow = rel_err(ref, oracle_outputs(case, setting, params, ip, iv), case["dt"])
            ok = ep <= args.tol and ev <= args.tol + allow
            if ok:
                matching.append(setting)
            label = ", ".join(f"{k}={v}" for k, v in setting.items()) or "-"
            tight = ok and ep <= 1e-6 and ev <= 1e-6 + allow
            print(f"{name:30s} {label:46s} {ep:9.2e} {ev:9.2e}  {('MATCH (also at 1e-6)' if tight else 'MATCH') if ok else 'differs'}"
                  + (f"   [reference: {ref['raised'][:60]}]" if "raised" in ref else ""))
        if not matching:
            all_ok = False
            print(f"{name:30s} -> NO setting of {names or 'the oracle'} reproduces the reference at {args.tol:g}")
        for s in names:
            votes[s] &= {m[s] for m in matching}
    print()
    for s, (_, cands) in SWITCHES.items():
        left = votes[s]
        touched = any(s in cases[n]["switches"] for n in refs)
        if not touched:
            print(f"[pin] {s:20s}: not exercised by the cases present")
        elif len(left) >= 1:
            shipped = cands[0]
            chosen = shipped if shipped in left else sorted(left)[0]
            both = " (every candidate matches: the cases cannot tell them apart)" if len(left) > 1 else ""
            print(f"[pin] {s:20s}: reference behaves as '{chosen}'{both}"
                  + ("" if chosen == shipped else f"   <-- the shipped default is '{shipped}': change it (include/mpk.h, "
                                                   f"oracle/mp_oracle.py, fancy_gym_amd/mp)"))
            all_ok = all_ok and chosen == shipped
        else:
            print(f"[pin] {s:20s}: no single setting matches every case")
            all_ok = False
    print("[pin] " + ("PINNED: the oracle with its shipped defaults reproduces the reference on every case"
                      if all_ok else "NOT pinned with the shipped defaults (see above)"))
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
