"""
Type-string factories of the black-box path (one table-driven module; the per-kind modules next to it only re-export
for import-path compatibility with ``fancy_gym.black_box.factory.*``).

Same strings, same exceptions as the reference:
  phase       'linear' | 'exp'            ('rhythmic', 'smooth' -> NotImplementedError)   phase_generator_factory.py:9-23
  basis       'rbf' | 'zero_rbf' | 'prodmp' ('rhythmic' -> NotImplementedError; prodmp needs the exp phase)  basis_generator_factory.py:8-23
  trajectory  'promp' | 'dmp' | 'prodmp'  (prodmp needs the prodmp basis)                 trajectory_generator_factory.py:7-21
  controller  'motor' | 'velocity' | 'position' | 'metaworld'                             controller_factory.py:9-21
Unknown strings raise ValueError naming the supported ones.
"""
from __future__ import annotations

from .. import controller as _ctrl
from ... import mp as _mp

PHASE_TYPES = ["linear", "exp", "rhythmic", "smooth"]
BASIS_TYPES = ["rbf", "zero_rbf", "rhythmic"]
TRAJECTORY_TYPES = ["promp", "dmp", "idmp"]
CONTROLLER_TYPES = ["motor", "velocity", "position", "metaworld"]

_PHASE = {"linear": _mp.LinearPhaseGenerator, "exp": _mp.ExpDecayPhaseGenerator}
_BASIS = {"rbf": _mp.NormalizedRBFBasisGenerator, "zero_rbf": _mp.ZeroPaddingNormalizedRBFBasisGenerator,
          "prodmp": _mp.ProDMPBasisGenerator}
_TRAJ = {"promp": _mp.ProMP, "dmp": _mp.DMP, "prodmp": _mp.ProDMP}
_CONTROLLER = {"motor": _ctrl.PDController, "velocity": _ctrl.VelController, "position": _ctrl.PosController,
               "metaworld": _ctrl.MetaWorldController}
_DECLARED_ONLY = {"phase": {"rhythmic", "smooth"}, "basis": {"rhythmic"}}


def _pick(kind: str, label: str, table: dict, listed: list, type_string: str):
    key = type_string.lower()
    if key in table:
        return table[key]
    if key in _DECLARED_ONLY.get(kind, ()):
        raise NotImplementedError()      # declared upstream, never implemented
    raise ValueError(f"Specified {label} type {key} not supported, please choose one of {listed}.")


def get_phase_generator(phase_generator_type, **kwargs):
    return _pick("phase", "phase generator", _PHASE, PHASE_TYPES, phase_generator_type)(**kwargs)


def get_basis_generator(basis_generator_type: str, phase_generator, **kwargs):
    cls = _pick("basis", "basis generator", _BASIS, BASIS_TYPES, basis_generator_type)
    if cls is _mp.ProDMPBasisGenerator:
        assert isinstance(phase_generator, _mp.ExpDecayPhaseGenerator)
    return cls(phase_generator, **kwargs)


def get_trajectory_generator(trajectory_generator_type: str, action_dim: int, basis_generator, **kwargs):
    cls = _pick("trajectory", "movement primitive", _TRAJ, TRAJECTORY_TYPES, trajectory_generator_type)
    if cls is _mp.ProDMP:
        assert isinstance(basis_generator, _mp.ProDMPBasisGenerator)
    return cls(basis_generator, action_dim, **kwargs)


def get_controller(controller_type: str, **kwargs):
    return _pick("controller", "controller", _CONTROLLER, CONTROLLER_TYPES, controller_type)(**kwargs)
