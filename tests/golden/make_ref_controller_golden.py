#!/usr/bin/env python3
"""
Generates tests/golden/ref_controllers.npz FROM THE REFERENCE'S OWN CODE -- run in the BUILD container only
(/root/reference does not exist on the GPU box; the fixture is data and travels, the reference's files do not).

The four tracking-controller files under /root/reference/fancy_gym/black_box/controller/ are the only arithmetic of the
hot path that lives in the reference checkout (they import nothing but numpy / typing and each other;
everything else of the path is mp_pytorch).  `fancy_gym/__init__.py` cannot be executed here (gymnasium is absent), so
the three package levels are registered as EMPTY namespace modules whose __path__ points into the checkout, and the
controller modules are then loaded by importlib UNMODIFIED, from where they lie.  Nothing of their text is copied.

What is stored (inputs + the reference's outputs, float64 as numpy promotes them):
  per BASELINE shape (cfg1 .. cfg5): seeded float32 desired positions / velocities [B, T, D], float64 state [B, D],
      the gains and action bounds of the reference's configuration, and
        pd / pos / vel          controller(des[b, t], state[b]) for every (b, t), one call per step as
                                BlackBoxWrapper.step makes them (black_box_wrapper.py:176-177), frozen state
        *_clip                  the same after np.clip(., low, high)  (black_box_wrapper.py:178-179)
        loop_act / loop_q / loop_qd
                                the per-step loop with the reference PDController IN the loop and the torque double
                                integrator restated beside it (base_reacher_torque.py:25-26: vel += dt * a; pos += dt * vel
                                -- that file imports gymnasium and cannot be loaded; two lines, restated HERE and said so)
  the reference's own known-answer grid (test/test_controller.py:14-73: 3-vectors of zeros / ones / arange against
      scalar and vector gains; metaworld with gripper 0 / 1 / 0.5), outputs of the reference classes
  metaworld on seeded 4-vectors (xyz + gripper)
  sha256 of each reference file that was loaded
"""
from __future__ import annotations

import hashlib
import importlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
PKG = os.path.join(REF, "fancy_gym", "black_box", "controller")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_controllers.npz")
FILES = ("base_controller.py", "pd_controller.py", "pos_controller.py", "vel_controller.py", "meta_world_controller.py")


def load_reference_controllers():
    """namespace modules for the three package levels (no __init__ of the reference runs), then the files as they lie"""
    for name, path in (("fancy_gym", os.path.join(REF, "fancy_gym")),
                       ("fancy_gym.black_box", os.path.join(REF, "fancy_gym", "black_box")),
                       ("fancy_gym.black_box.controller", PKG)):
        assert name not in sys.modules, f"{name} is already imported -- run this script in a fresh interpreter"
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    mods = {f[:-3]: importlib.import_module("fancy_gym.black_box.controller." + f[:-3]) for f in FILES}
    for f, m in zip(FILES, mods.values()):
        assert os.path.realpath(m.__file__) == os.path.realpath(os.path.join(PKG, f)), m.__file__
    return (mods["pd_controller"].PDController, mods["pos_controller"].PosController,
            mods["vel_controller"].VelController, mods["meta_world_controller"].MetaWorldController)


# gains / bounds of the five BASELINE configurations, as the reference's files hold them
SHAPES = {
    # fancy_ProMP/Reacher5d-v0: _BB_DEFAULTS['ProMP'] controller_kwargs (envs/registry.py:80-83), Box(-1, 1) actions
    "cfg1": dict(D=5, T=200, B=3, p=1.0, d=0.1, lo=-1.0, hi=1.0, dt=0.02),
    # BoxPushingDense: envs/mujoco/box_pushing/mp_wrapper.py:12-13 vector gains, box_pushing_env.py:65 Box(-1, 1)
    "cfg2": dict(D=7, T=100, B=5, p=0.01 * np.array([120., 120., 120., 120., 50., 30., 10.]),
                 d=0.01 * np.array([10., 10., 10., 10., 6., 5., 3.]), lo=-1.0, hi=1.0, dt=0.02),
    # fancy_DMP/Reacher7d: _BB_DEFAULTS['DMP'] (envs/registry.py:100-103)
    "cfg3": dict(D=7, T=200, B=3, p=1.0, d=0.1, lo=-1.0, hi=1.0, dt=0.02),
    # BoxPushingDenseReplan: the same gains (box_pushing/mp_wrapper.py:64-93 inherits :12-13)
    "cfg4": dict(D=7, T=100, B=4, p=0.01 * np.array([120., 120., 120., 120., 50., 30., 10.]),
                 d=0.01 * np.array([10., 10., 10., 10., 6., 5., 3.]), lo=-1.0, hi=1.0, dt=0.02),
    # TableTennis4D: envs/mujoco/table_tennis/mp_wrapper.py:19-20, table_tennis_env.py:97 Box(-1, 1) float32
    "cfg5": dict(D=7, T=350, B=2, p=0.5 * np.array([1.0, 4.0, 2.0, 4.0, 1.0, 4.0, 1.0]),
                 d=0.5 * np.array([0.1, 0.4, 0.2, 0.4, 0.1, 0.4, 0.1]), lo=-1.0, hi=1.0, dt=0.008),
}


def main():
    PD, Pos, Vel, Meta = load_reference_controllers()
    out = {}
    for k, (name, s) in enumerate(sorted(SHAPES.items())):
        rng = np.random.default_rng(500 + k)
        B, T, D = s["B"], s["T"], s["D"]
        # smooth-ish desired trajectories of O(1) size, float32 as get_trajectory returns them (black_box_wrapper.py:117-118)
        des_pos = np.cumsum(rng.standard_normal((B, T, D)) * 0.05, axis=1).astype(np.float32)
        des_vel = (rng.standard_normal((B, T, D)) * 1.5).astype(np.float32)
        q0 = rng.uniform(-1, 1, (B, D))                 # float64: MuJoCo / numpy state
        qd0 = rng.uniform(-0.5, 0.5, (B, D))
        lo = np.full(D, s["lo"], np.float32)            # gymnasium Box bounds are float32 arrays
        hi = np.full(D, s["hi"], np.float32)
        pd, pos_c, vel_c = PD(p_gains=s["p"], d_gains=s["d"]), Pos(), Vel()
        res = {n: np.empty((B, T, D), np.float64) for n in ("pd", "pos", "vel")}
        for b in range(B):
            for t in range(T):
                res["pd"][b, t] = pd.get_action(des_pos[b, t], des_vel[b, t], q0[b], qd0[b])
                res["pos"][b, t] = pos_c.get_action(des_pos[b, t], des_vel[b, t], q0[b], qd0[b])
                res["vel"][b, t] = vel_c.get_action(des_pos[b, t], des_vel[b, t], q0[b], qd0[b])
        # the per-step loop with the reference controller in it; the plant's two lines restated (see the docstring)
        loop_act = np.empty((B, T, D), np.float64)
        loop_q, loop_qd = np.empty((B, D)), np.empty((B, D))
        for b in range(B):
            q, qd = q0[b].copy(), qd0[b].copy()
            for t in range(T):
                a = np.clip(pd.get_action(des_pos[b, t], des_vel[b, t], q, qd), lo, hi)
                qd = qd + s["dt"] * a
                q = q + s["dt"] * qd
                loop_act[b, t] = a
            loop_q[b], loop_qd[b] = q, qd
        out.update({f"{name}_des_pos": des_pos, f"{name}_des_vel": des_vel, f"{name}_q0": q0, f"{name}_qd0": qd0,
                    f"{name}_p_gains": np.asarray(s["p"], np.float64), f"{name}_d_gains": np.asarray(s["d"], np.float64),
                    f"{name}_lo": lo, f"{name}_hi": hi, f"{name}_dt": np.float64(s["dt"]),
                    f"{name}_loop_act": loop_act, f"{name}_loop_q": loop_q, f"{name}_loop_qd": loop_qd})
        # pos / vel controllers return their input object (pos_controller.py:8-9): only the clipped image is stored
        assert np.array_equal(res["pos"], des_pos) and np.array_equal(res["vel"], des_vel)
        out[f"{name}_pd"] = res["pd"]
        for n, v in res.items():
            out[f"{name}_{n}_clip"] = np.clip(v, lo, hi)

    # the reference's known-answer grid, test/test_controller.py:14-44 (outputs of the reference classes)
    vecs = [np.zeros(3), np.ones(3), np.arange(0, 3)]
    gains = [0, 1, 0.5, np.zeros(3), np.ones(3), np.arange(0, 3)]
    grid_in, grid_out = [], []
    for position in vecs:
        for velocity in vecs:
            for cp in vecs:
                for cv in vecs:
                    for pg in gains:
                        for dg in gains:
                            a = PD(p_gains=pg, d_gains=dg)(position, velocity, cp, cv)
                            grid_in.append(np.concatenate([position, velocity, cp, cv, np.broadcast_to(pg, 3),
                                                           np.broadcast_to(dg, 3)]).astype(np.float64))
                            grid_out.append(np.asarray(a, np.float64))
    out["grid_in"], out["grid_pd"] = np.stack(grid_in), np.stack(grid_out)

    # metaworld: test/test_controller.py:57-73 grid + seeded 4-vectors (xyz + gripper opening)
    mw = Meta()
    mw_in, mw_out = [], []
    for position in vecs:
        for cp in vecs:
            for g in (0, 1, 0.5):
                dp, c = np.append(position, g).astype(np.float64), np.append(cp, -1.0)
                mw_in.append(np.concatenate([dp, c]))
                mw_out.append(mw(dp, np.zeros(4), c, np.zeros(4)))
    rng = np.random.default_rng(77)
    for _ in range(64):
        dp, c = rng.standard_normal(4).astype(np.float32), rng.uniform(-1, 1, 4)
        mw_in.append(np.concatenate([dp.astype(np.float64), c]))
        mw_out.append(mw.get_action(dp, np.zeros(4, np.float32), c, np.zeros(4)))
    out["metaworld_in"], out["metaworld_out"] = np.stack(mw_in), np.stack(mw_out).astype(np.float64)

    # shape errors the reference raises (pd_controller.py:22-27)
    raised = []
    for pv in ((np.ones(3), np.ones(4)), (np.ones(4), np.ones(3)), (np.ones(4), np.ones(4))):
        try:
            PD()(pv[0], pv[1], np.ones(3), np.ones(3))
            raised.append("")
        except Exception as e:      # noqa: BLE001 -- the type is the datum
            raised.append(type(e).__name__)
    sha = {f: hashlib.sha256(open(os.path.join(PKG, f), "rb").read()).hexdigest() for f in FILES}
    meta = ("generated from /root/reference controller/*.py (fancy_gym/black_box/controller: the reference's own "
            "PDController / PosController / VelController / MetaWorldController, loaded unmodified through namespace "
            "modules; fancy_gym/__init__.py never executed).  loop_* additionally uses the torque double integrator of "
            "base_reacher_torque.py:25-26 RESTATED in the generator (that file needs gymnasium).  "
            f"numpy {np.__version__}; python {sys.version.split()[0]}; shape errors raised: {raised}; "
            "generator sha256 " + hashlib.sha256(open(os.path.abspath(__file__), "rb").read()).hexdigest() + "; "
            + "; ".join(f"{f} sha256 {h}" for f, h in sha.items()))
    out["meta"] = np.array(meta)
    out["shape_errors"] = np.array(raised)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")
    print(meta)


if __name__ == "__main__":
    main()
