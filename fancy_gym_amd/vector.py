"""
VectorBlackBox -- host-side bridge for environments that cannot live on the GPU (MuJoCo & co.):
N ordinary ``BlackBoxWrapper`` episodes whose plans are generated in ONE kernel launch, then tracked on their host envs
(``BlackBoxWrapper.step_planned``), optionally in a thread pool (MuJoCo releases the GIL while stepping).

Every per-episode rule of the reference's wrapper (clipping, frozen tau / delay, replanning init_time, conditioning on
the desired state, validity hooks, info collation) is applied by the individual wrappers -- this class only batches the
``traj_gen.get_traj_pos / get_traj_vel`` part of ``black_box_wrapper.py:96-120``.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np
import torch

from .black_box.black_box_wrapper import BlackBoxWrapper


class VectorBlackBox:

    def __init__(self, envs: Sequence[BlackBoxWrapper], num_workers: int = 0):
        """envs: BlackBoxWrappers of ONE movement-primitive configuration (e.g. ``[make_bb(...) for _ in range(N)]``)"""
        if not envs:
            raise ValueError("need at least one environment")
        self.envs: List[BlackBoxWrapper] = list(envs)
        first = self.envs[0]
        if first.learn_sub_trajectories:
            raise ValueError("sub-trajectory mode plans a different horizon per episode and cannot be batched")
        self.engine = first.traj_gen.engine()
        for e in self.envs[1:]:
            same = (e.traj_gen.num_params == first.traj_gen.num_params and e.traj_gen.mp_type == first.traj_gen.mp_type
                    and e.duration == first.duration and e.dt == first.dt)
            if not same:
                raise ValueError("all environments must share one movement-primitive configuration")
        self.num_envs = len(self.envs)
        self.action_space = first.action_space
        self.observation_space = first.observation_space
        self._pool = ThreadPoolExecutor(num_workers) if num_workers > 0 else None

    def reset(self, *, seed: Optional[int] = None, options=None):
        outs = [e.reset(seed=None if seed is None else seed + i, options=options) for i, e in enumerate(self.envs)]
        return np.stack([o for o, _ in outs]), [i for _, i in outs]

    def get_trajectories(self, actions: np.ndarray):
        """plans for all envs: ONE launch, one device-to-host copy; returns float32 arrays [N, T, D]"""
        actions = np.asarray(actions)
        assert actions.shape[0] == self.num_envs
        full, it, ip, iv = [], [], [], []
        for env, a in zip(self.envs, actions):
            env._stage_plan(a)                     # clip, freeze tau / delay, boundary condition, init_time
            g = env.traj_gen
            full.append(g._full_params()); it.append(g.init_time)
            ip.append(g.init_pos); iv.append(g.init_vel)
        cfg = self.engine.config
        shared = not (cfg.learn_tau or cfg.learn_delay) and len(set(it)) == 1
        init_time = float(it[0]) if shared else torch.tensor(np.asarray(it, np.float32), device=self.engine.device)
        pos, vel = self.engine.trajectory(np.stack(full), np.stack(ip), np.stack(iv), init_time)
        if not shared and self.engine.mp_type == "prodmp":
            # per-episode phase: a scaled time beyond the pre-computed table range only shows on the device; raise what
            # every individual wrapper (and mp_pytorch) raises -- RuntimeError("Time is beyond the pre-computation range")
            self.engine.check_range()
        return pos.cpu().numpy(), vel.cpu().numpy()

    def step(self, actions: np.ndarray):
        pos, vel = self.get_trajectories(actions)
        jobs = [(e, a, p, v) for e, a, p, v in zip(self.envs, actions, pos, vel)]
        run = lambda j: j[0].step_planned(j[1], j[2], j[3])  # noqa: E731
        results = list(self._pool.map(run, jobs)) if self._pool else [run(j) for j in jobs]
        obs, rew, term, trunc, infos = zip(*results)
        return (np.stack(obs), np.asarray(rew, dtype=np.float64), np.asarray(term, dtype=bool),
                np.asarray(trunc, dtype=bool), list(infos))

    def close(self):
        if self._pool:
            self._pool.shutdown()
        for e in self.envs:
            e.close()
