class BaseController:
    """Tracking-controller plugin: ``get_action(des_pos, des_vel, c_pos, c_vel)`` (reference base_controller.py:1-7).

    ``device_type`` names the controller in the HIP rollout kernels (``None`` = host only)."""

    device_type = None

    def get_action(self, des_pos, des_vel, c_pos, c_vel):
        raise NotImplementedError

    def __call__(self, des_pos, des_vel, c_pos, c_vel):
        return self.get_action(des_pos, des_vel, c_pos, c_vel)
