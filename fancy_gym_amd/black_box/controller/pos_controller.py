from .base_controller import BaseController


class PosController(BaseController):
    """action = desired position (reference pos_controller.py:8-9)"""

    device_type = "position"

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        return des_pos
