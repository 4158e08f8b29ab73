from .basis_generator_factory import get_basis_generator  # noqa: F401
from .controller_factory import get_controller  # noqa: F401
from .phase_generator_factory import get_phase_generator  # noqa: F401
from .trajectory_generator_factory import get_trajectory_generator  # noqa: F401
