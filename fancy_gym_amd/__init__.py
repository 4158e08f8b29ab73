"""
fancy_gym_amd -- MI355X-native drop-in for fancy_gym's black-box movement-primitive hot path
(MP parameter vector -> (pos, vel) trajectory -> per-step tracking-controller action).

All arithmetic of the path runs in hand-written HIP kernels (csrc/, gfx950) behind the C-ABI of include/mpk.h;
this package is the Python host mirror of the reference's interfaces for that path (same names as
``fancy_gym/__init__.py:1-20`` where they exist there).
"""
from ._lib import MPKLibraryError  # noqa: F401
from .engine import RolloutSpec, TrajectoryEngine  # noqa: F401
from .black_box.black_box_wrapper import BlackBoxWrapper  # noqa: F401
from .black_box.raw_interface_wrapper import RawInterfaceWrapper  # noqa: F401
from .batched import BatchedBlackBox  # noqa: F401
from .vector import VectorBlackBox  # noqa: F401
from .envs.registry import (ALL_MOVEMENT_PRIMITIVE_ENVIRONMENTS, MOVEMENT_PRIMITIVE_ENVIRONMENTS_FOR_NS,  # noqa: F401
                            register, upgrade)
from .utils.make_env_helpers import make_bb  # noqa: F401

__version__ = "0.1.0"


def make(*args, **kwargs):
    raise Exception("fancy_gym_amd.make is not a thing: register your env and use the gymnasium-style make of "
                    "fancy_gym_amd._gym (the reference's fancy_gym.make raises as well).")
