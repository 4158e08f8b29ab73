"""type string -> basis generator (reference factory/basis_generator_factory.py:8-23)"""
from ...mp import (ExpDecayPhaseGenerator, NormalizedRBFBasisGenerator, PhaseGenerator, ProDMPBasisGenerator,
                   ZeroPaddingNormalizedRBFBasisGenerator)

ALL_TYPES = ["rbf", "zero_rbf", "rhythmic"]


def get_basis_generator(basis_generator_type: str, phase_generator: PhaseGenerator, **kwargs):
    key = basis_generator_type.lower()
    if key == "rbf":
        return NormalizedRBFBasisGenerator(phase_generator, **kwargs)
    if key == "zero_rbf":
        return ZeroPaddingNormalizedRBFBasisGenerator(phase_generator, **kwargs)
    if key == "prodmp":
        assert isinstance(phase_generator, ExpDecayPhaseGenerator)
        return ProDMPBasisGenerator(phase_generator, **kwargs)
    if key == "rhythmic":
        raise NotImplementedError()
    raise ValueError(f"Specified basis generator type {key} not supported, please choose one of {ALL_TYPES}.")
