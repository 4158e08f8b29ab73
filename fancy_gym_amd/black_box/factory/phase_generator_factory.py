from .factories import PHASE_TYPES as ALL_TYPES, get_phase_generator  # noqa: F401  (import-path alias)
