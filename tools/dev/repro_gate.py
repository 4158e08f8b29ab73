#!/usr/bin/env python3
"""replay one case of tests/test_gpu_fuzz.py::test_random_batched_episode_with_validity_gate and print the configuration:
    python tools/dev/repro_gate.py SEED"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_fuzz as F
from oracle import mp_oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(321_000 + seed)
pc, bc, tc, dt, dur, B, _ = F.random_case(rng)
print(pc); print(bc); print(tc); print("dt", dt, "dur", dur, "B", min(B, 16), "T", round(dur / dt))
import pytest
from fancy_gym_amd import _lib
try:
    F.test_random_batched_episode_with_validity_gate.__wrapped__  # noqa
except AttributeError:
    pass
def opt(k, v):
    print("option", k, v); _lib.set_option(k, int(v))
try:
    F.test_random_batched_episode_with_validity_gate(seed, opt)
    print("passed")
except AssertionError as e:
    print("FAILED:", str(e)[:400])
