"""facade: mp_pytorch.phase_gn (factory/phase_generator_factory.py:9-23)"""
import numpy as np


class PhaseGenerator:
    kind = None

    def __init__(self, tau=3.0, delay=0.0, learn_tau=False, learn_delay=False, **kwargs):
        self.tau, self.delay, self.learn_tau, self.learn_delay = float(tau), float(delay), bool(learn_tau), bool(learn_delay)
        self.tau_bound = list(kwargs.pop("tau_bound", [1e-5, np.inf]))
        self.delay_bound = list(kwargs.pop("delay_bound", [0.0, np.inf]))
        self.alpha_phase = float(kwargs.pop("alpha_phase", 3.0))
        if kwargs:
            raise TypeError(f"unexpected phase kwargs {sorted(kwargs)}")


class LinearPhaseGenerator(PhaseGenerator):
    kind = "linear"


class ExpDecayPhaseGenerator(PhaseGenerator):
    kind = "exp"
