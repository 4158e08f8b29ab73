from .factories import TRAJECTORY_TYPES as ALL_TYPES, get_trajectory_generator  # noqa: F401  (import-path alias)
