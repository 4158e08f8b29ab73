"""
Tracking controllers of the black-box path, host side (one module; the per-controller modules next to it only re-export
for import-path compatibility with ``fancy_gym.black_box.controller.*``).

A controller maps (desired position, desired velocity, current position, current velocity) to a raw action of the
wrapped step-based env.  ``device_type`` names the same law inside the HIP kernels (``tile_epilogue``, ``k_pd_rollout``,
the closed-loop section of ``k_traj_stream``); ``None`` means the law only exists on the host.

Reference: fancy_gym/black_box/controller/{base,pd,pos,vel,meta_world}_controller.py.
"""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np

Gains = Union[float, Tuple, np.ndarray]


def _same_shape(what: str, desired, current) -> None:
    if desired.shape != current.shape:
        raise ValueError(f"Mismatch in dimension between desired {what} {desired.shape} and current {what} "
                         f"{current.shape}")


class BaseController:
    """interface: ``get_action(des_pos, des_vel, c_pos, c_vel)``; instances are callable (base_controller.py:1-7)"""
    device_type = None

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        raise NotImplementedError

    def __call__(self, *state):
        return self.get_action(*state)


class PDController(BaseController):
    """torque = P (q_des - q) + D (qdot_des - qdot), gains scalar or per DoF (pd_controller.py:21-29)"""
    device_type = "motor"

    def __init__(self, p_gains: Gains = 1, d_gains: Gains = 0.5):
        self.p_gains, self.d_gains = p_gains, d_gains

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        _same_shape("position", des_pos, c_pos)
        _same_shape("velocity", des_vel, c_vel)
        pos_err, vel_err = des_pos - c_pos, des_vel - c_vel
        return self.p_gains * pos_err + self.d_gains * vel_err


class PosController(BaseController):
    """position-controlled envs: the action IS the desired position (pos_controller.py:8-9)"""
    device_type = "position"

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        return des_pos


class VelController(BaseController):
    """velocity-controlled envs: the action IS the desired velocity (vel_controller.py:8-9)"""
    device_type = "velocity"

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        return des_vel


class MetaWorldController(BaseController):
    """metaworld end-effector control: xyz as a position delta, last entry = raw gripper opening
    (meta_world_controller.py:15-25).  There is no batchable metaworld plant, so on the device it exists for a frozen state
    only (``RolloutSpec("metaworld", D, plant="static")`` with ``TrajectoryEngine.trajectory_actions`` / ``pd_rollout``: it
    runs on the motor kernels with unit position gains); environments are stepped on the host (``VectorBlackBox``)"""

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        *_, gripper = des_pos
        target_xyz, current_xyz = des_pos[:-1], c_pos[:-1]
        _same_shape("position", target_xyz, current_xyz)
        return np.hstack([target_xyz - current_xyz, gripper])
