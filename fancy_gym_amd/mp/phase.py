"""Phase generators: configuration + the learn_tau / learn_delay parameter bookkeeping."""
from __future__ import annotations

import numpy as np
import torch

_INF = float("inf")


class PhaseGenerator:
    """
    Owns tau / delay, whether they are learned, and their bounds.  ``set_params`` consumes ``[tau?, delay?]`` from the
    front of a parameter vector; once set they stay frozen ("finalized") until ``reset()`` -- this is what makes a
    learned delay act on the first plan of a replanning episode only (reference test/test_replanning_sequencing.py:
    231-282).  ``tau_bound`` / ``delay_bound`` exist only when the quantity is learned
    (reference black_box_wrapper.py:60-65 probes them with ``hasattr``).
    """
    type_name = "abstract"

    def __init__(self, tau: float = 3.0, delay: float = 0.0, learn_tau: bool = False, learn_delay: bool = False,
                 **kwargs):
        self._tau0, self._delay0 = float(tau), float(delay)
        self._tau, self._delay = float(tau), float(delay)
        self.learn_tau, self.learn_delay = bool(learn_tau), bool(learn_delay)
        if self.learn_tau:
            self.tau_bound = list(kwargs.get("tau_bound", [1e-5, _INF]))
            assert len(self.tau_bound) == 2
        if self.learn_delay:
            self.delay_bound = list(kwargs.get("delay_bound", [0.0, _INF]))
            assert len(self.delay_bound) == 2
        self.is_finalized = False

    # mp_pytorch exposes tensors (the reference's tests call ``env.traj_gen.tau.numpy()``)
    @property
    def tau(self) -> torch.Tensor:
        return torch.tensor(self._tau, dtype=torch.float32)

    @property
    def delay(self) -> torch.Tensor:
        return torch.tensor(self._delay, dtype=torch.float32)

    @property
    def num_params(self) -> int:
        return int(self.learn_tau) + int(self.learn_delay)

    def set_params(self, params: np.ndarray) -> np.ndarray:
        i = 0
        if self.learn_tau:
            tau = float(params[..., i])
            assert tau > 0, "tau must be positive"
            if not self.is_finalized:
                self._tau = float(np.float32(tau))
            i += 1
        if self.learn_delay:
            delay = float(params[..., i])
            assert delay >= 0, "delay must be non-negative"
            if not self.is_finalized:
                self._delay = float(np.float32(delay))
            i += 1
        self.finalize()
        return params[..., i:]

    def get_params_bounds(self) -> np.ndarray:
        lo, hi = [], []
        if self.learn_tau:
            lo.append(self.tau_bound[0]); hi.append(self.tau_bound[1])
        if self.learn_delay:
            lo.append(self.delay_bound[0]); hi.append(self.delay_bound[1])
        return np.array([lo, hi], dtype=np.float32).reshape(2, -1)

    def finalize(self):
        self.is_finalized = True

    def reset(self):
        self.is_finalized = False
        self._tau, self._delay = self._tau0, self._delay0


class LinearPhaseGenerator(PhaseGenerator):
    """s = clip((t - delay) / tau, 0, 1)   (factory/phase_generator_factory.py:11-12: 'linear')"""
    type_name = "linear"


class ExpDecayPhaseGenerator(LinearPhaseGenerator):
    """x = exp(-alpha_phase * max((t - delay) / tau, 0))   (factory/phase_generator_factory.py:13-14: 'exp')"""
    type_name = "exp"

    def __init__(self, tau: float = 3.0, delay: float = 0.0, alpha_phase: float = 3.0, learn_tau: bool = False,
                 learn_delay: bool = False, **kwargs):
        super().__init__(tau=tau, delay=delay, learn_tau=learn_tau, learn_delay=learn_delay, **kwargs)
        self.alpha_phase = float(alpha_phase)
