"""Host-data helpers (reference: ``src/evox/utils/io.py:6-26``): external simulators hand back
float64 / int64 numpy data; device state is kept in 32-bit."""
from __future__ import annotations

from functools import wraps

import numpy as np
import torch

from ..core.tree import tree_map


def to_x32_if_needed(values):
    def cast(x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if isinstance(x, torch.Tensor):
            if x.dtype == torch.float64:
                return x.to(torch.float32)
        return x

    return tree_map(cast, values)


def x32_func_call(func):
    @wraps(func)
    def inner(*args, **kwargs):
        return to_x32_if_needed(func(*args, **kwargs))

    return inner
