from collections.abc import Mapping, MutableMapping

import numpy as np


def get_numpy(x):
    """tensor-like -> ndarray: the device-to-host copy point of the path (reference utils/utils.py:27-36)."""
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()


def nested_update(base: MutableMapping, update):
    """plain recursive dict update (reference utils/utils.py:39-50)"""
    for k, v in update.items():
        base[k] = nested_update(base.get(k, {}), v) if isinstance(v, Mapping) else v
    return base
