"""Problem interface (reference: ``src/evox/core/problem.py:9-31``)."""
from __future__ import annotations

from typing import Any, Tuple

from .module import Stateful
from .state import State


class Problem(Stateful):
    """``evaluate(state, pop) -> (fitness, state)``: fitness is ``(n,)`` for one objective and
    ``(n, m)`` for m objectives; lower is better."""

    def evaluate(self, state: State, pop: Any) -> Tuple[Any, State]:
        raise NotImplementedError
