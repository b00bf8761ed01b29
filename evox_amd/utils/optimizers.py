"""Gradient-based optimizers used by the evolution-strategy family.

A small functional API in the spirit of optax (which the reference relies on but which is not
part of this stack): ``opt = adam(lr)``; ``state = opt.init(params)``; ``updates, state =
opt.update(grads, state, params)``; ``params = apply_updates(params, updates)``.  Parameters may
be any tree of tensors.  ``OptaxWrapper`` makes such an optimizer a ``Stateful`` module whose
state lives in the State tree (reference: ``src/evox/utils/common.py:142-153``), and ``ClipUp``
is the PGPE optimizer of Toklu et al. (reference: ``es_variants/pgpe.py:34-64``).
"""
from __future__ import annotations

from typing import Any, Callable, NamedTuple, Optional

import torch

from ..core import tree as _tree
from ..core.module import Stateful
from ..core.state import State
from ..core.struct import dataclass, pytree_field


class GradientTransformation(NamedTuple):
    init: Callable
    update: Callable


def apply_updates(params, updates):
    return _tree.tree_map(lambda p, u: p + u, params, updates)


def sgd(learning_rate: float, momentum: Optional[float] = None) -> GradientTransformation:
    def init(params):
        if momentum is None:
            return {}
        return {"trace": _tree.tree_map(torch.zeros_like, params)}

    def update(grads, state, params=None):
        if momentum is None:
            return _tree.tree_map(lambda g: -learning_rate * g, grads), state
        trace = _tree.tree_map(lambda t, g: momentum * t + g, state["trace"], grads)
        return _tree.tree_map(lambda t: -learning_rate * t, trace), {"trace": trace}

    return GradientTransformation(init, update)


def adam(learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> GradientTransformation:
    def init(params):
        first = _tree.tree_leaves(params)[0]
        return {"count": torch.zeros((), dtype=torch.float32, device=first.device),
                "mu": _tree.tree_map(torch.zeros_like, params), "nu": _tree.tree_map(torch.zeros_like, params)}

    def update(grads, state, params=None):
        count = state["count"] + 1
        mu = _tree.tree_map(lambda m, g: b1 * m + (1 - b1) * g, state["mu"], grads)
        nu = _tree.tree_map(lambda v, g: b2 * v + (1 - b2) * g * g, state["nu"], grads)
        c1 = 1 - b1 ** count
        c2 = 1 - b2 ** count
        updates = _tree.tree_map(lambda m, v: -learning_rate * (m / c1) / (torch.sqrt(v / c2) + eps), mu, nu)
        return updates, {"count": count, "mu": mu, "nu": nu}

    return GradientTransformation(init, update)


def rmsprop(learning_rate: float, decay: float = 0.9, eps: float = 1e-8) -> GradientTransformation:
    def init(params):
        return {"nu": _tree.tree_map(torch.zeros_like, params)}

    def update(grads, state, params=None):
        nu = _tree.tree_map(lambda v, g: decay * v + (1 - decay) * g * g, state["nu"], grads)
        return _tree.tree_map(lambda g, v: -learning_rate * g / (torch.sqrt(v) + eps), grads, nu), {"nu": nu}

    return GradientTransformation(init, update)


_BY_NAME = {"adam": adam, "sgd": sgd, "rmsprop": rmsprop}


def by_name(name: str, learning_rate: float) -> GradientTransformation:
    if name not in _BY_NAME:
        raise ValueError(f"Unknown optimizer {name}")
    return _BY_NAME[name](learning_rate=learning_rate)


@dataclass
class OptaxWrapper(Stateful):
    """A functional optimizer as a module: its state is the ``opt_state`` field of its State."""

    optimizer: Any = pytree_field(static=True)
    init_params: Any = None

    def setup(self, key):
        return State(opt_state=self.optimizer.init(self.init_params))

    def update(self, state, grads, params=None):
        updates, opt_state = self.optimizer.update(grads, state.opt_state, params)
        return updates, state.replace(opt_state=opt_state)


def tree_l2_norm(tree):
    return torch.sqrt(sum((leaf ** 2).sum() for leaf in _tree.tree_leaves(tree)))


@dataclass
class ClipUp(Stateful):
    """Normalised-gradient momentum with a velocity norm cap."""

    step_size: float
    max_speed: float
    momentum: float
    params: Any = None

    def setup(self, key):
        return State(velocity=_tree.tree_map(torch.zeros_like, self.params))

    def update(self, state, gradient, _params=None):
        gnorm = tree_l2_norm(gradient)
        velocity = _tree.tree_map(lambda v, g: self.momentum * v + self.step_size * g / gnorm, state.velocity, gradient)
        vnorm = tree_l2_norm(velocity)
        scale = torch.where(vnorm > self.max_speed, self.max_speed / vnorm, torch.ones_like(vnorm))
        velocity = _tree.tree_map(lambda v: v * scale, velocity)
        return _tree.tree_map(lambda v: -v, velocity), state.replace(velocity=velocity)
