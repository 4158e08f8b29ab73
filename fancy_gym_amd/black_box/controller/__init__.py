from .controllers import (BaseController, MetaWorldController, PDController, PosController,  # noqa: F401
                          VelController)
