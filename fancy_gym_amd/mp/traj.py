"""
Trajectory generators with the MPInterface surface ``BlackBoxWrapper`` calls
(reference black_box_wrapper.py:57,62-65,102,106,113-118,124-125,226):
``set_duration, set_params, set_initial_conditions, get_traj_pos, get_traj_vel, get_params_bounds, reset`` and the
attributes ``phase_gn, basis_gn, learn_tau, tau, num_params``.

The arithmetic is NOT here: ``get_traj_pos / get_traj_vel`` launch the HIP kernels through ``TrajectoryEngine``
(created lazily, so configuration objects can be built and inspected on a machine without a GPU).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .basis import BasisGenerator, ProDMPBasisGenerator, ZeroPaddingNormalizedRBFBasisGenerator


class MPInterface:
    mp_type = "abstract"

    def __init__(self, basis_gn: BasisGenerator, num_dof: int, weights_scale: float = 1.0, device=None, **kwargs):
        self.basis_gn = basis_gn
        self.phase_gn = basis_gn.phase_generator
        self.num_dof = int(num_dof)
        self.weights_scale = float(weights_scale)
        self._device = device
        self._engine = None
        self._engine_key = None
        self.duration: Optional[float] = None
        self.dt: Optional[float] = None
        self.reset()

    # ---- configuration -------------------------------------------------------------------------------------------
    @property
    def learn_tau(self) -> bool:
        return self.phase_gn.learn_tau

    @property
    def learn_delay(self) -> bool:
        return self.phase_gn.learn_delay

    @property
    def tau(self) -> torch.Tensor:
        return self.phase_gn.tau

    @property
    def num_basis(self) -> int:
        return self.basis_gn.num_basis

    @property
    def _num_local_params(self) -> int:
        return self.num_basis * self.num_dof

    @property
    def num_params(self) -> int:
        return self._num_local_params + self.basis_gn.num_params

    def get_params_bounds(self) -> torch.Tensor:
        """[2, P]: tau / delay bounds, everything else +-inf (reference black_box_wrapper.py:122-127 unpacks the two
        rows and calls ``.numpy()`` on each)."""
        pb = self.basis_gn.get_params_bounds()
        local = np.empty((2, self._num_local_params), np.float32)
        local[0], local[1] = -np.inf, np.inf
        return torch.from_numpy(np.concatenate([pb, local], axis=1).astype(np.float32))

    def _engine_extra(self) -> dict:
        return {}

    def engine(self):
        """The TrajectoryEngine for the current (duration, dt); built on first use."""
        if self.duration is None or self.dt is None:
            raise RuntimeError("set_duration(duration, dt) must be called before a trajectory can be generated")
        from ..engine import TrajectoryEngine
        if self._engine is None:
            pg = self.phase_gn
            kw = dict(mp_type=self.mp_type, phase_type=pg.type_name, num_dof=self.num_dof, dt=self.dt,
                      duration=self.duration, tau=pg._tau0, delay=pg._delay0,
                      alpha_phase=getattr(pg, "alpha_phase", 3.0), learn_tau=pg.learn_tau, learn_delay=pg.learn_delay,
                      tau_bound=getattr(pg, "tau_bound", (1e-5, float("inf"))),
                      delay_bound=getattr(pg, "delay_bound", (0.0, float("inf"))),
                      weights_scale=self.weights_scale, device=self._device)
            kw.update(self.basis_gn.engine_kwargs())
            kw.update(self._engine_extra())
            self._engine = TrajectoryEngine(**kw)
        else:
            self._engine.set_duration(self.duration, self.dt)
        return self._engine

    # ---- per-episode state ---------------------------------------------------------------------------------------
    def reset(self):
        self.basis_gn.reset()
        self.params = None
        self.init_time = None
        self.init_pos = None
        self.init_vel = None
        self._clear()

    def _clear(self):
        self._pos = None
        self._vel = None

    def set_duration(self, duration: Optional[float], dt: float, include_init_time: bool = False):
        if include_init_time:
            raise NotImplementedError("include_init_time=True is not used by the black-box path")
        dt = float(dt)
        if duration is None:
            # sub-trajectory mode (reference black_box_wrapper.py:98-102): the plan lasts round(tau / dt) steps
            duration = round(float(self.phase_gn._tau) / dt) * dt
        self.duration, self.dt = float(duration), dt
        self._clear()

    def set_params(self, params) -> np.ndarray:
        params = np.asarray(params, dtype=np.float32)
        assert params.shape[-1] == self.num_params, \
            f"expected {self.num_params} parameters, got {params.shape[-1]}"
        rest = self.basis_gn.set_params(params)
        self.params = np.ascontiguousarray(rest[..., :self._num_local_params])
        self._clear()
        return rest[..., self._num_local_params:]

    def set_initial_conditions(self, init_time, init_pos, init_vel, **_ignored):
        self.init_time = float(np.asarray(init_time, dtype=np.float32))
        self.init_pos = np.asarray(init_pos, dtype=np.float32).reshape(-1)
        self.init_vel = np.asarray(init_vel, dtype=np.float32).reshape(-1)
        self._clear()

    def _full_params(self) -> np.ndarray:
        """[tau?, delay?, local...] with the (possibly frozen) phase parameters in front."""
        head = []
        if self.phase_gn.learn_tau:
            head.append(self.phase_gn._tau)
        if self.phase_gn.learn_delay:
            head.append(self.phase_gn._delay)
        return np.concatenate([np.asarray(head, np.float32), self.params.reshape(-1)]).astype(np.float32)

    def _compute(self):
        assert self.params is not None, "set_params() first"
        D = self.num_dof
        ip = self.init_pos if self.init_pos is not None else np.zeros(D, np.float32)
        iv = self.init_vel if self.init_vel is not None else np.zeros(D, np.float32)
        it = self.init_time if self.init_time is not None else 0.0
        eng = self.engine()
        if D == 0:
            z = torch.zeros((eng.num_steps, 0), dtype=torch.float32)
            self._pos, self._vel = z, z.clone()
            return
        # one episode: pinned staging in, one launch, (pos | vel) back in one copy -> CPU tensors, which is also what
        # mp_pytorch hands to fancy_gym's get_numpy on its default device
        self._pos, self._vel = eng.trajectory_host(self._full_params(), ip, iv, it)

    def show_scaled_basis(self, plot: bool = False):
        """
        mp_pytorch's inspection helper the reference's examples call (fancy_gym/examples/mp_params_tuning.py:7): the basis
        functions times their parameter scale on 1000 times from ``delay - tau`` to ``delay + 2 tau``, evaluated by the
        device row functions (mpk_scaled_basis).  Returns ``(times [1000], basis [1000, K])`` as numpy arrays; ``plot=True``
        additionally draws them with matplotlib as upstream does.
        """
        tau, delay = float(self.phase_gn._tau0), float(self.phase_gn._delay0)
        times = np.linspace(delay - tau, delay + 2 * tau, 1000, dtype=np.float32)
        if self.duration is None:
            self.set_duration(tau, 0.01)          # any grid: the basis is evaluated at `times`, not on the plan's grid
        basis = self.engine().scaled_basis(times)
        if plot:  # pragma: no cover - needs a display
            import matplotlib.pyplot as plt
            plt.figure()
            for i in range(basis.shape[-1]):
                plt.plot(times, basis[:, i], label=f"w_basis_{i}")
            plt.grid(); plt.legend()
            plt.axvline(x=delay, linestyle="--", color="k", alpha=0.3)
            plt.axvline(x=delay + tau, linestyle="--", color="k", alpha=0.3)
            plt.show()
        return times, basis

    def get_traj_pos(self, **_ignored) -> torch.Tensor:
        if self._pos is None:
            self._compute()
        return self._pos

    def get_traj_vel(self, **_ignored) -> torch.Tensor:
        if self._vel is None:
            self._compute()
        return self._vel


class ProMP(MPInterface):
    """'promp' (factory/trajectory_generator_factory.py:11-12)"""
    mp_type = "promp"

    def __init__(self, basis_gn, num_dof, weights_scale: float = 1.0, **kwargs):
        super().__init__(basis_gn, num_dof, weights_scale, **kwargs)
        self.has_zero_padding = isinstance(basis_gn, ZeroPaddingNormalizedRBFBasisGenerator)


class DMP(MPInterface):
    """'dmp' (factory/trajectory_generator_factory.py:13-14): one extra goal parameter per DoF"""
    mp_type = "dmp"

    def __init__(self, basis_gn, num_dof, weights_scale: float = 1.0, goal_scale: float = 1.0, alpha: float = 25,
                 **kwargs):
        self.goal_scale, self.alpha = float(goal_scale), float(alpha)
        # how the first returned sample relates to the initial condition: 'init' (default) | 'step' (include/mpk.h
        # MPK_DMP_FIRST_*; SURVEY A.6 "(?)")
        self.dmp_first_sample = kwargs.pop("dmp_first_sample", "init")
        super().__init__(basis_gn, num_dof, weights_scale, **kwargs)

    @property
    def _num_local_params(self) -> int:
        return (self.num_basis + 1) * self.num_dof

    def _engine_extra(self) -> dict:
        return dict(goal_scale=self.goal_scale, dmp_alpha=self.alpha, dmp_first_sample=self.dmp_first_sample)


class ProDMP(MPInterface):
    """'prodmp' (factory/trajectory_generator_factory.py:15-18)"""
    mp_type = "prodmp"

    def __init__(self, basis_gn, num_dof, weights_scale: float = 1.0, goal_scale: float = 1.0, **kwargs):
        assert isinstance(basis_gn, ProDMPBasisGenerator)
        self.goal_scale = float(goal_scale)
        self.auto_scale_basis = bool(kwargs.pop("auto_scale_basis", False))
        self.relative_goal = bool(kwargs.pop("relative_goal", False))
        self.disable_weights = bool(kwargs.pop("disable_weights", False))
        self.disable_goal = bool(kwargs.pop("disable_goal", False))
        # the reference passes goal_offset in box_pushing/mp_wrapper.py:77 and table_tennis/mp_wrapper.py:114; in
        # mp_pytorch <= 0.1.3 it is believed to disappear into **kwargs (SURVEY A.5 (?)): goal_offset_mode 'ignore'
        # (default) accepts and drops it, 'add' adds it to the goal (include/mpk.h MPK_GOAL_OFFSET_*)
        self.goal_offset = kwargs.pop("goal_offset", None)
        self.goal_offset_mode = kwargs.pop("goal_offset_mode", "ignore")
        # relative_goal: init_pos joins the raw goal parameter ('before_scale', default since ABI 3) or the scaled goal
        # ('after_scale') -- include/mpk.h MPK_RELGOAL_*; unpinned either way
        self.relative_goal_mode = kwargs.pop("relative_goal_mode", "before_scale")
        kwargs.pop("duration", None)   # _BB_DEFAULTS['ProDMP'] carries a stray 'duration' (registry.py:108)
        super().__init__(basis_gn, num_dof, weights_scale, **kwargs)

    @property
    def _num_local_params(self) -> int:
        k = (0 if self.disable_weights else self.num_basis) + (0 if self.disable_goal else 1)
        return k * self.num_dof

    def _engine_extra(self) -> dict:
        return dict(goal_scale=self.goal_scale, auto_scale_basis=self.auto_scale_basis,
                    relative_goal=self.relative_goal, disable_goal=self.disable_goal,
                    disable_weights=self.disable_weights, relative_goal_mode=self.relative_goal_mode,
                    goal_offset_mode=self.goal_offset_mode, goal_offset=float(self.goal_offset or 0.0))
