"""Algorithm interface: ask for candidates, be told their fitness.

Reference: ``src/evox/core/algorithm.py:10-96``."""
from __future__ import annotations

from typing import Any, Tuple

from .module import Stateful
from .state import State


class Algorithm(Stateful):
    """Base class of optimisation algorithms.

    ``ask(state) -> (candidates, state)`` proposes solutions, ``tell(state, fitness) -> state``
    consumes their (minimisation) fitness.  Algorithms whose first generation differs (e.g. the
    initial population must be evaluated before any variation) additionally define
    ``init_ask`` / ``init_tell``; the workflow uses them for the first step only."""

    def ask(self, state: State) -> Tuple[Any, State]:
        raise NotImplementedError

    def tell(self, state: State, fitness) -> State:
        raise NotImplementedError


def _overrides(obj, name: str) -> bool:
    fn = getattr(type(obj), name, None)
    return callable(fn) and fn is not getattr(Algorithm, name, None)


def has_init_ask(algorithm) -> bool:
    return _overrides(algorithm, "init_ask")


def has_init_tell(algorithm) -> bool:
    return _overrides(algorithm, "init_tell")
