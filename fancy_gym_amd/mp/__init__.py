"""
Host-side mirrors of the mp_pytorch classes the reference instantiates through its factories
(fancy_gym/black_box/factory/*.py).  They hold configuration and per-episode state only; every trajectory number is
computed by the HIP kernels behind ``TrajectoryEngine``.
"""
from .phase import ExpDecayPhaseGenerator, LinearPhaseGenerator, PhaseGenerator  # noqa: F401
from .basis import (BasisGenerator, NormalizedRBFBasisGenerator, ProDMPBasisGenerator,  # noqa: F401
                    ZeroPaddingNormalizedRBFBasisGenerator)
from .traj import DMP, MPInterface, ProDMP, ProMP  # noqa: F401
