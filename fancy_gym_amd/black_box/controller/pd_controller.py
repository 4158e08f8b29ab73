from typing import Tuple, Union

from .base_controller import BaseController


class PDController(BaseController):
    """trq = p_gains * (des_pos - c_pos) + d_gains * (des_vel - c_vel)   (reference pd_controller.py:21-29).

    Host version for a single env stepped from Python; the batched paths run the same formula in float64 inside
    ``k_traj_tiles`` / ``k_traj_stream`` / ``k_pd_rollout``."""

    device_type = "motor"

    def __init__(self, p_gains: Union[float, Tuple] = 1, d_gains: Union[float, Tuple] = 0.5):
        self.p_gains = p_gains
        self.d_gains = d_gains

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        for what, des, cur in (("position", des_pos, c_pos), ("velocity", des_vel, c_vel)):
            if des.shape != cur.shape:
                raise ValueError(f"Mismatch in dimension between desired {what} {des.shape} and current {what} "
                                 f"{cur.shape}")
        return self.p_gains * (des_pos - c_pos) + self.d_gains * (des_vel - c_vel)
