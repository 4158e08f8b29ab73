from .factories import BASIS_TYPES as ALL_TYPES, get_basis_generator  # noqa: F401  (import-path alias)
