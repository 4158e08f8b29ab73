"""
fancy_gym_amd -- MI355X-native drop-in for fancy_gym's black-box movement-primitive hot path
(MP parameter vector -> (pos, vel) trajectory -> per-step tracking-controller action).

All arithmetic of the path runs in hand-written HIP kernels (csrc/, gfx950) behind the C-ABI of include/mpk.h;
this package is the Python host mirror of the reference's interfaces for that path.
"""
from ._lib import MPKLibraryError  # noqa: F401
from .engine import RolloutSpec, TrajectoryEngine  # noqa: F401

__version__ = "0.1.0"
