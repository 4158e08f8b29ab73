"""Basis generators: configuration only -- evaluation happens in k_build_shared / k_traj_rows."""
from __future__ import annotations

from .phase import ExpDecayPhaseGenerator, PhaseGenerator


class BasisGenerator:
    type_name = "abstract"

    def __init__(self, phase_generator: PhaseGenerator, num_basis: int = 10):
        self.phase_generator = phase_generator
        self._num_basis = int(num_basis)

    @property
    def num_basis(self) -> int:
        """learnable basis functions per DoF"""
        return self._num_basis

    # parameter bookkeeping is delegated to the phase generator (mp_pytorch's BasisGenerator does the same)
    @property
    def num_params(self) -> int:
        return self.phase_generator.num_params

    def set_params(self, params):
        return self.phase_generator.set_params(params)

    def get_params_bounds(self):
        return self.phase_generator.get_params_bounds()

    def reset(self):
        self.phase_generator.reset()

    def engine_kwargs(self) -> dict:
        raise NotImplementedError


class NormalizedRBFBasisGenerator(BasisGenerator):
    """'rbf' (factory/basis_generator_factory.py:10-11)"""
    type_name = "rbf"

    def __init__(self, phase_generator, num_basis: int = 10, basis_bandwidth_factor: float = 3,
                 num_basis_outside: int = 0, single_rbf_mode: str = "unit_gap", **_ignored):
        super().__init__(phase_generator, num_basis)
        self.basis_bandwidth_factor = float(basis_bandwidth_factor)
        self.num_basis_outside = int(num_basis_outside)
        # ONE basis function in total has no neighbouring centre: 'unit_gap' (default) | 'refuse' (include/mpk.h)
        self.single_rbf_mode = single_rbf_mode

    def engine_kwargs(self) -> dict:
        return dict(basis_type="rbf", num_basis=self.num_basis, basis_bandwidth_factor=self.basis_bandwidth_factor,
                    num_basis_outside=self.num_basis_outside, single_rbf_mode=self.single_rbf_mode)


class ZeroPaddingNormalizedRBFBasisGenerator(NormalizedRBFBasisGenerator):
    """'zero_rbf' (factory/basis_generator_factory.py:12-13): extra, non-learnable RBFs at the start / goal whose
    weights are zero, so the trajectory starts (ends) at the offset position."""
    type_name = "zero_rbf"

    def __init__(self, phase_generator, num_basis: int = 10, num_basis_zero_start: int = 2,
                 num_basis_zero_goal: int = 0, basis_bandwidth_factor: float = 3, single_rbf_mode: str = "unit_gap",
                 **_ignored):
        super().__init__(phase_generator, num_basis, basis_bandwidth_factor, 0, single_rbf_mode)
        self.num_basis_zero_start = int(num_basis_zero_start)
        self.num_basis_zero_goal = int(num_basis_zero_goal)

    def engine_kwargs(self) -> dict:
        return dict(basis_type="zero_rbf", num_basis=self.num_basis,
                    basis_bandwidth_factor=self.basis_bandwidth_factor,
                    num_basis_zero_start=self.num_basis_zero_start, num_basis_zero_goal=self.num_basis_zero_goal,
                    single_rbf_mode=self.single_rbf_mode)


class ProDMPBasisGenerator(NormalizedRBFBasisGenerator):
    """'prodmp' (factory/basis_generator_factory.py:14-17): pre-computed integral-form tables on a scaled-time grid."""
    type_name = "prodmp"

    def __init__(self, phase_generator, num_basis: int = 10, basis_bandwidth_factor: float = 3,
                 num_basis_outside: int = 0, dt: float = 0.01, alpha: float = 25, pre_compute_length_factor: int = 6,
                 **_ignored):
        assert isinstance(phase_generator, ExpDecayPhaseGenerator)
        assert pre_compute_length_factor <= 6, "For numerical stability, please use a length factor <= 6."
        super().__init__(phase_generator, num_basis, basis_bandwidth_factor, num_basis_outside)
        self.dt, self.alpha = float(dt), float(alpha)
        self.pre_compute_length_factor = int(pre_compute_length_factor)

    def engine_kwargs(self) -> dict:
        return dict(basis_type="prodmp", num_basis=self.num_basis, basis_bandwidth_factor=self.basis_bandwidth_factor,
                    num_basis_outside=self.num_basis_outside, basis_alpha=self.alpha, basis_dt=self.dt,
                    pre_compute_length_factor=self.pre_compute_length_factor)
