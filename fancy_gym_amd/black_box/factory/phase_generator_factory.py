"""type string -> phase generator (reference factory/phase_generator_factory.py:9-23: same strings, same errors)"""
from ...mp import ExpDecayPhaseGenerator, LinearPhaseGenerator

ALL_TYPES = ["linear", "exp", "rhythmic", "smooth"]
_IMPLEMENTED = {"linear": LinearPhaseGenerator, "exp": ExpDecayPhaseGenerator}


def get_phase_generator(phase_generator_type, **kwargs):
    key = phase_generator_type.lower()
    if key in _IMPLEMENTED:
        return _IMPLEMENTED[key](**kwargs)
    if key in ALL_TYPES:            # 'rhythmic', 'smooth': declared upstream, never implemented
        raise NotImplementedError()
    raise ValueError(f"Specified phase generator type {key} not supported, please choose one of {ALL_TYPES}.")
