from .controllers import MetaWorldController  # noqa: F401  (import-path alias)
