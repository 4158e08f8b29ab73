from .controllers import BaseController  # noqa: F401  (import-path alias)
