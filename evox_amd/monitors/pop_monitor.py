"""Population monitor: snapshots the algorithm's population / fitness after every step.
API parity: reference ``src/evox/monitors/pop_monitor.py:13-116``.  Snapshots stay on the
device they were produced on (state tensors are immutable, so keeping a reference is free);
``to_host=True`` moves them to host memory instead."""
from __future__ import annotations

import warnings

from ..core.monitor import Monitor


class PopMonitor(Monitor):
    def __init__(self, population_name="population", fitness_name="fitness", fitness_only=False, to_host=False):
        super().__init__()
        self.population_name = population_name
        self.fitness_name = fitness_name
        self.fitness_only = fitness_only
        self.to_host = to_host
        self.population_history = []
        self.fitness_history = []

    def hooks(self):
        return ["post_step"]

    def _keep(self, t):
        return t.detach().cpu() if self.to_host else t.detach()

    def post_step(self, state):
        algo = state.get_child_state("algorithm")
        if not self.fitness_only:
            self.population_history.append(self._keep(getattr(algo, self.population_name)))
        self.fitness_history.append(self._keep(getattr(algo, self.fitness_name)))

    def plot(self, problem_pf=None, **kwargs):
        from ..vis_tools import plot
        if not self.fitness_history:
            warnings.warn("No fitness history recorded, return None")
            return None
        n_objs = 1 if self.fitness_history[0].dim() == 1 else self.fitness_history[0].shape[1]
        if n_objs == 1:
            return plot.plot_obj_space_1d(self.fitness_history, **kwargs)
        if n_objs == 2:
            return plot.plot_obj_space_2d(self.fitness_history, problem_pf, **kwargs)
        if n_objs == 3:
            return plot.plot_obj_space_3d(self.fitness_history, problem_pf=problem_pf, **kwargs)
        warnings.warn("Not supported yet.")
        return None

    def get_latest_fitness(self):
        return self.fitness_history[-1]

    def get_latest_population(self):
        return self.population_history[-1]

    def get_population_history(self):
        return self.population_history

    def get_fitness_history(self):
        return self.fitness_history
