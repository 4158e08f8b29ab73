import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "timing: asserts on a measured duration (wall clock, HIP events, a bench line's rates); "
                                       "collected LAST, behind every test that compares the HIP path with the oracle or a fixture")
    # the suites bind libmpk.so through ctypes: (re)build it if the checkout has none or the sources are newer
    import __graft_entry__ as entry
    if entry._stale():
        entry.build()


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # Wall-clock assertions go to the end of the session: under `pytest -x` a noisy box that fails one of them must not leave a
    # single parity test uncollected behind it.  Stable: both groups keep their file order.
    items[:] = [i for i in items if "timing" not in i.keywords] + [i for i in items if "timing" in i.keywords]
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def mpk_option():
    """
    Kernel-selection overrides for the tests that pin every kernel variant: ``mpk_option("quad", 2)`` sets the process-
    wide default of libmpk (mpk_set_option with a NULL handle, include/mpk.h); everything is automatic again after the
    test.  Values may be given as the strings the parametrisations carry.
    """
    from fancy_gym_amd import _lib

    def setter(key, value):
        _lib.set_option(key, int(value))
    yield setter
    _lib.reset_options()
