"""type string -> trajectory generator (reference factory/trajectory_generator_factory.py:7-21)"""
from ...mp import DMP, BasisGenerator, ProDMP, ProDMPBasisGenerator, ProMP

ALL_TYPES = ["promp", "dmp", "idmp"]


def get_trajectory_generator(trajectory_generator_type: str, action_dim: int, basis_generator: BasisGenerator,
                             **kwargs):
    key = trajectory_generator_type.lower()
    if key == "promp":
        return ProMP(basis_generator, action_dim, **kwargs)
    if key == "dmp":
        return DMP(basis_generator, action_dim, **kwargs)
    if key == "prodmp":
        assert isinstance(basis_generator, ProDMPBasisGenerator)
        return ProDMP(basis_generator, action_dim, **kwargs)
    raise ValueError(f"Specified movement primitive type {key} not supported, please choose one of {ALL_TYPES}.")
