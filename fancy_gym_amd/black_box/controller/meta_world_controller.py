import numpy as np

from .base_controller import BaseController


class MetaWorldController(BaseController):
    """xyz position delta + raw gripper opening (reference meta_world_controller.py:15-25); host only."""

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        xyz_des, xyz_cur = des_pos[:-1], c_pos[:-1]
        if xyz_des.shape != xyz_cur.shape:
            raise ValueError(f"Mismatch in dimension between desired position {xyz_des.shape} and current position "
                             f"{xyz_cur.shape}")
        return np.hstack([xyz_des - xyz_cur, des_pos[-1]])
