from .controllers import VelController  # noqa: F401  (import-path alias)
