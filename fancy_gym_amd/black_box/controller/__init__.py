from .base_controller import BaseController  # noqa: F401
from .meta_world_controller import MetaWorldController  # noqa: F401
from .pd_controller import PDController  # noqa: F401
from .pos_controller import PosController  # noqa: F401
from .vel_controller import VelController  # noqa: F401
