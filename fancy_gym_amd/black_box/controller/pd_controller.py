from .controllers import PDController  # noqa: F401  (import-path alias)
