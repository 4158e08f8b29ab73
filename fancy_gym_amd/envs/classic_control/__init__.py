from .simple_reacher import SimpleReacherEnv, SimpleReacherMPWrapper  # noqa: F401
