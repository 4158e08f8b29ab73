from .controllers import PosController  # noqa: F401  (import-path alias)
