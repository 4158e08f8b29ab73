"""facade: mp_pytorch.basis_gn (factory/basis_generator_factory.py:10-17)"""


class BasisGenerator:
    kind = None

    def __init__(self, phase_generator, num_basis=10, basis_bandwidth_factor=3, num_basis_outside=0, **kw):
        self.phase_generator = phase_generator
        self.kw = dict(num_basis=int(num_basis), basis_bandwidth_factor=float(basis_bandwidth_factor),
                       num_basis_outside=int(num_basis_outside), **kw)


class NormalizedRBFBasisGenerator(BasisGenerator):
    kind = "rbf"


class ZeroPaddingNormalizedRBFBasisGenerator(BasisGenerator):
    kind = "zero_rbf"

    def __init__(self, phase_generator, num_basis=10, num_basis_zero_start=2, num_basis_zero_goal=0,
                 basis_bandwidth_factor=3, **kw):
        super().__init__(phase_generator, num_basis, basis_bandwidth_factor, 0,
                         num_basis_zero_start=int(num_basis_zero_start), num_basis_zero_goal=int(num_basis_zero_goal), **kw)


class ProDMPBasisGenerator(BasisGenerator):
    kind = "prodmp"

    def __init__(self, phase_generator, num_basis=10, basis_bandwidth_factor=3, num_basis_outside=0, dt=0.01, alpha=25,
                 pre_compute_length_factor=6, **kw):
        super().__init__(phase_generator, num_basis, basis_bandwidth_factor, num_basis_outside, dt=float(dt),
                         alpha=float(alpha), pre_compute_length_factor=int(pre_compute_length_factor), **kw)
