from .factories import (get_basis_generator, get_controller, get_phase_generator,  # noqa: F401
                        get_trajectory_generator)
