"""Workflow marker base (reference: ``src/evox/core/workflow.py:4-5``)."""
from .module import Stateful


class Workflow(Stateful):
    """A driver that owns an algorithm and a problem and advances them one step at a time."""

    def step(self, state):
        raise NotImplementedError
