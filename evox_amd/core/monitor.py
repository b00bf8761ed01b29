"""Monitor hook base (reference: ``src/evox/core/monitor.py:1-47``).

A workflow calls the hooks a monitor lists in :meth:`hooks`, in this order per step::

    pre_step(state) · pre_ask(state) · post_ask(state, cand) · pre_eval(state, cand, tcand)
    post_eval(state, cand, tcand, fitness) · pre_tell(state, cand, tcand, fitness, tfitness)
    post_tell(state) · post_step(state)
"""
from __future__ import annotations

HOOK_NAMES = ("pre_step", "pre_ask", "post_ask", "pre_eval", "post_eval", "pre_tell", "post_tell", "post_step")


class Monitor:
    def set_opt_direction(self, opt_direction):
        self.opt_direction = opt_direction

    def hooks(self):
        """Names of the hooks this monitor implements."""
        return []

    def pre_step(self, state):
        pass

    def pre_ask(self, state):
        pass

    def post_ask(self, state, cand_sol):
        pass

    def pre_eval(self, state, cand_sol, transformed_cand_sol):
        pass

    def post_eval(self, state, cand_sol, transformed_cand_sol, fitness):
        pass

    def pre_tell(self, state, cand_sol, transformed_cand_sol, fitness, transformed_fitness):
        pass

    def post_tell(self, state):
        pass

    def post_step(self, state):
        pass

    def flush(self):
        """Wait for asynchronous device→host transfers this monitor started."""
        pass

    def close(self):
        self.flush()
