#!/usr/bin/env python3
"""Single-episode latency of the drop-in path (what a fancy_gym user sees at B = 1): python tools/b1_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fancy_gym_amd  # noqa: E402,F401
from fancy_gym_amd import _gym  # noqa: E402


def main():
    for env_id in ("fancy_ProDMP/LongSimpleReacher-v0", "fancy_ProMP/LongSimpleReacher-v0", "fancy_DMP/LongSimpleReacher-v0"):
        env = _gym.make(env_id)
        env.reset(seed=0)
        a = env.action_space.sample()
        for _ in range(20):
            env.get_trajectory(a)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            env.get_trajectory(a)
        t_plan = (time.perf_counter() - t0) / n
        env.reset(seed=0)
        env.step(a)
        t0 = time.perf_counter()
        m = 10
        for _ in range(m):
            env.reset(seed=0)
            env.step(a)
        t_step = (time.perf_counter() - t0) / m
        print(f"{env_id}: get_trajectory {t_plan * 1e6:.0f} us; reset + step (200 host env steps) {t_step * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
