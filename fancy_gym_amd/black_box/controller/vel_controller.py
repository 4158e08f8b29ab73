from .base_controller import BaseController


class VelController(BaseController):
    """action = desired velocity (reference vel_controller.py:8-9)"""

    device_type = "velocity"

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        return des_vel
