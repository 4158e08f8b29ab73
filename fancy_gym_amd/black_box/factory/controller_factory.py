"""type string -> tracking controller (reference factory/controller_factory.py:9-21)"""
from ..controller import MetaWorldController, PDController, PosController, VelController

ALL_TYPES = ["motor", "velocity", "position", "metaworld"]
_TABLE = {"motor": PDController, "velocity": VelController, "position": PosController,
          "metaworld": MetaWorldController}


def get_controller(controller_type: str, **kwargs):
    key = controller_type.lower()
    if key not in _TABLE:
        raise ValueError(f"Specified controller type {key} not supported, please choose one of {ALL_TYPES}.")
    return _TABLE[key](**kwargs)
