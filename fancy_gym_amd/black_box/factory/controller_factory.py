from .factories import CONTROLLER_TYPES as ALL_TYPES, get_controller  # noqa: F401  (import-path alias)
