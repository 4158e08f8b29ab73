"""
RawInterfaceWrapper -- the plugin interface a step-based environment exposes so that it can be driven by a movement
primitive.  Mirror of ``fancy_gym/black_box/raw_interface_wrapper.py`` (same names, argument meaning and defaults).
"""
from __future__ import annotations

from typing import Tuple, Union

import numpy as np

from .._gym import Wrapper


class RawInterfaceWrapper(Wrapper):
    """Subclass and provide at least ``current_pos`` and ``current_vel`` (reference: raw_interface_wrapper.py:24-44)."""

    @property
    def context_mask(self) -> np.ndarray:
        """bool[obs_dim]: which observation entries form the contextual observation (reference :10-22)."""
        return np.ones(self.env.observation_space.shape[0], dtype=bool)

    @property
    def current_pos(self) -> Union[float, int, np.ndarray, Tuple]:
        """position of the controlled DoFs; dimension = action dimension (reference :24-33)"""
        raise NotImplementedError

    @property
    def current_vel(self) -> Union[float, int, np.ndarray, Tuple]:
        """velocity of the controlled DoFs; dimension = action dimension (reference :35-44)"""
        raise NotImplementedError

    @property
    def dt(self) -> float:
        """control period of the wrapped env (reference :46-53)"""
        return self.env.dt

    def preprocessing_and_validity_callback(self, action: np.ndarray, pos_traj: np.ndarray, vel_traj: np.ndarray,
                                            tau_bound: list = None, delay_bound: list = None
                                            ) -> Tuple[bool, np.ndarray, np.ndarray]:
        """(valid, pos_traj, vel_traj): hook to validate / post-process the desired trajectory (reference :55-72)."""
        return True, pos_traj, vel_traj

    def set_episode_arguments(self, action, pos_traj, vel_traj):
        """deprecated predecessor of the validity callback; identity by default (reference :74-87)"""
        return pos_traj, vel_traj

    def episode_callback(self, action: np.ndarray, pos_traj: np.ndarray, vel_traj: np.ndarray) -> Tuple[bool]:
        """hook for envs whose action carries non-MP entries (reference :89-101)"""
        return True

    def invalid_traj_callback(self, action: np.ndarray, pos_traj: np.ndarray, vel_traj: np.ndarray,
                              return_contextual_obs: bool = True, tau_bound: list = None, delay_bound: list = None
                              ) -> Tuple[np.ndarray, float, bool, bool, dict]:
        """
        Artificial (obs, reward, terminated, truncated, info) for an invalid trajectory.  The reference *calls* this
        with six arguments (black_box_wrapper.py:170-171) although its base class declares five (:103-104); the six
        argument form is the one kept here.
        """
        return np.zeros(1), 0, True, False, {}
