"""Small tensor utilities shared by algorithms, operators and metrics.

API parity with the reference's ``src/evox/utils/common.py:15-310``; implementations are
batched torch expressions (pairwise kernels go through ``evox_amd.ops`` when on a GPU).
"""
from __future__ import annotations

import math
import pickle
from collections.abc import Iterable
from typing import Callable, List, Sequence, Union

import torch

from ..core import tree as _tree
from ..core.module import Stateful
from ..core.state import State
from ..core.struct import dataclass, pytree_field


def min_by(values, keys):
    """The entry of ``values`` with the smallest key (lists are concatenated first)."""
    if isinstance(values, (list, tuple)):
        values = torch.cat(list(values))
        keys = torch.cat(list(keys))
    i = torch.argmin(keys)
    return values[i], keys[i]


def pairwise_func(x: torch.Tensor, y: torch.Tensor, func: Callable) -> torch.Tensor:
    """``out[i, j] = func(x[i], y[j])`` for a func written for two vectors; evaluated by
    broadcasting ``x[:, None]`` against ``y[None]`` with the reduction on the last axis."""
    return torch.vmap(lambda a: torch.vmap(lambda b: func(a, b))(y))(x)


def euclidean_dist(x, y):
    return torch.linalg.vector_norm(x - y, dim=-1) if x.dim() > 0 else (x - y).abs()


def manhattan_dist(x, y):
    return (x - y).abs().sum(-1)


def chebyshev_dist(x, y):
    return (x - y).abs().amax(-1)


def pairwise_euclidean_dist(x, y):
    from ..ops.distance import cdist

    return cdist(x, y)


def pairwise_manhattan_dist(x, y):
    from ..ops.distance import cdist

    return cdist(x, y, p=1.0)


def pairwise_chebyshev_dist(x, y):
    from ..ops.distance import cdist

    return cdist(x, y, p=float("inf"))


def pair_max(a, b):
    return (a - b).amax(-1)


def cal_max(x, y):
    """``out[i, j] = max_k (x[i, k] - y[j, k])`` (additive ε-indicator matrix)."""
    return (x[:, None, :] - y[None, :, :]).amax(-1)


def cos_dist(x, y):
    """Cosine similarity matrix between the rows of ``x`` and the rows of ``y``."""
    xn = x / torch.linalg.vector_norm(x, dim=-1, keepdim=True)
    yn = y / torch.linalg.vector_norm(y, dim=-1, keepdim=True)
    return xn @ yn.T


def _dominate(x, y):
    return (x <= y).all(-1) & (x < y).any(-1)


def dominate_relation(x, y):
    """Boolean matrix ``A[i, j]`` = ``x[i]`` Pareto-dominates ``y[j]``."""
    a, b = x[:, None, :], y[None, :, :]
    return (a <= b).all(-1) & (a < b).any(-1)


def new_dist_mat(xs):
    """Euclidean distance matrix of the rows of ``xs`` through the Gram matrix."""
    sq = (xs * xs).sum(-1)
    d2 = sq[:, None] + sq[None, :] - 2.0 * (xs @ xs.T)
    return d2.clamp_min(0).sqrt()


def compose(*functions):
    if len(functions) == 1 and isinstance(functions[0], Iterable):
        functions = tuple(functions[0])

    def composed(carry):
        for f in functions:
            carry = f(carry)
        return carry

    return composed


def rank(array):
    """0-based rank of every element of a 1-D tensor (ascending)."""
    order = torch.argsort(array, stable=True)
    r = torch.empty_like(order)
    r[order] = torch.arange(order.shape[0], device=array.device)
    return r


def rank_based_fitness(raw_fitness):
    """Centered ranks in ``[-0.5, 0.5]``."""
    n = raw_fitness.shape[0]
    return rank(raw_fitness).to(torch.float32) / (n - 1) - 0.5


class TreeAndVector:
    """Convert between a tree of tensors (e.g. network parameters) and flat vectors."""

    def __init__(self, dummy_input):
        leaves, self.treedef = _tree.tree_flatten(dummy_input)
        self.shapes = [tuple(x.shape) for x in leaves]
        self.slice_sizes = [int(math.prod(s)) for s in self.shapes]
        self.start_indices = [0]
        for s in self.slice_sizes[:-1]:
            self.start_indices.append(self.start_indices[-1] + s)
        self.num_params = sum(self.slice_sizes)

    def to_vector(self, x):
        return torch.cat([l.reshape(-1) for l in _tree.tree_leaves(x)], dim=0)

    def batched_to_vector(self, x):
        return torch.cat([l.reshape(l.shape[0], -1) for l in _tree.tree_leaves(x)], dim=1)

    def to_tree(self, x):
        leaves = [x[s:s + n].reshape(shape) for s, n, shape in zip(self.start_indices, self.slice_sizes, self.shapes)]
        return _tree.tree_unflatten(self.treedef, leaves)

    def batched_to_tree(self, x):
        b = x.shape[0]
        leaves = [x[:, s:s + n].reshape((b,) + shape)
                  for s, n, shape in zip(self.start_indices, self.slice_sizes, self.shapes)]
        return _tree.tree_unflatten(self.treedef, leaves)

    def __getstate__(self):
        return {"dummy_tree": _tree.tree_unflatten(self.treedef, list(self.start_indices)),
                "shapes": self.shapes, "start_indices": self.start_indices, "slice_sizes": self.slice_sizes}

    def __setstate__(self, d):
        _, self.treedef = _tree.tree_flatten(d["dummy_tree"])
        self.shapes, self.start_indices, self.slice_sizes = d["shapes"], d["start_indices"], d["slice_sizes"]
        self.num_params = sum(self.slice_sizes)


def parse_opt_direction(opt_direction: Union[str, Iterable]):
    """``"min"`` → 1, ``"max"`` → -1, a sequence → a float tensor of ±1."""
    def one(d):
        if d == "min":
            return 1
        if d == "max":
            return -1
        raise ValueError(f"opt_direction is either 'min' or 'max', got {d}")

    if isinstance(opt_direction, str):
        return one(opt_direction)
    if isinstance(opt_direction, Iterable):
        return torch.tensor([one(d) for d in opt_direction], dtype=torch.float32)
    raise ValueError(f"opt_direction should be a str or an iterable of str, got {type(opt_direction)}")


def frames2gif(frames, save_path, duration=0.1):
    try:
        import imageio
    except ImportError as e:  # pragma: no cover
        raise ImportError("imageio is required for rendering (not bundled with this image)") from e
    import numpy as np

    with imageio.get_writer(save_path, mode="I", duration=duration) as writer:
        for image in frames:
            writer.append_data(np.array(image, dtype=np.uint8))
    print("Gif saved to: ", save_path)


class AggregationFunction:
    """Scalarising functions of decomposition-based MOEAs: ``pbi``, ``tchebycheff``,
    ``tchebycheff_norm``, ``modified_tchebycheff``, ``weighted_sum``.  All take fitness ``f``
    ``(n, m)``, weights ``w`` ``(n, m)`` and the ideal point ``z``."""

    def __init__(self, function_name: str):
        table = {"pbi": self.pbi, "tchebycheff": self.tchebycheff, "tchebycheff_norm": self.tchebycheff_norm,
                 "modified_tchebycheff": self.modified_tchebycheff, "weighted_sum": self.weighted_sum}
        if function_name not in table:
            raise ValueError("Unsupported function")
        self.function = table[function_name]

    def pbi(self, f, w, z, *args):
        nw = torch.linalg.vector_norm(w, dim=-1)
        f = f - z
        d1 = (f * w).sum(-1) / nw
        d2 = torch.linalg.vector_norm(f - d1[..., None] * w / nw[..., None], dim=-1)
        return d1 + 5 * d2

    def tchebycheff(self, f, w, z, *args):
        return ((f - z).abs() * w).amax(-1)

    def tchebycheff_norm(self, f, w, z, z_max, *args):
        return ((f - z).abs() / (z_max - z) * w).amax(-1)

    def modified_tchebycheff(self, f, w, z, *args):
        return ((f - z).abs() / w).amax(-1)

    def weighted_sum(self, f, w, *args):
        return (f * w).sum(-1)

    def __call__(self, *args, **kwargs):
        return self.function(*args, **kwargs)
