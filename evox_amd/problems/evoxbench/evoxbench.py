"""Adapters for the EvoXBench neural-architecture-search suites.
Reference: ``src/evox/problems/evoxbench/evoxbench.py:12-75``.

EvoXBench evaluates architectures on the host from its own database; the adapter moves the
population to host memory, calls the benchmark under a seed derived from the problem's key
(its evaluators are noisy and use numpy's global generator) and returns float32 objectives on
the population's device.  Any object with ``evaluate(pop)``, ``evaluator.n_objs`` and
``search_space.lb/ub`` works, which is also how the adapter is tested without the package."""
from __future__ import annotations

import numpy as np
import torch

from ...core import random as rnd
from ...core.problem import Problem
from ...core.state import State


def evaluate_with_seed(benchmark, seed, pop):
    np.random.seed(seed)
    return np.asarray(benchmark.evaluate(pop)).astype(np.float32)


class EvoXBenchProblem(Problem):
    def __init__(self, benchmark):
        super().__init__()
        self.benchmark = benchmark
        self.n_objs = benchmark.evaluator.n_objs
        self.lb = benchmark.search_space.lb
        self.ub = benchmark.search_space.ub

    def setup(self, key):
        return State(key=key)

    def evaluate(self, state, pop):
        key, subkey = rnd.split(state.key)
        seed = int(rnd.randint(subkey, (1,), 0, 2 ** 31 - 1)[0])
        host = pop.detach().cpu().numpy()
        fit = evaluate_with_seed(self.benchmark, seed, host).reshape(host.shape[0], self.n_objs)
        return torch.as_tensor(fit).to(pop.device), state.replace(key=key)


def _suite(name, problem_id, upper):
    if not (isinstance(problem_id, int) and 1 <= problem_id <= upper):
        raise AssertionError(f"For {name}, problem_id must be an integer between 1 and {upper}")
    try:
        from evoxbench import test_suites
    except ImportError as e:
        raise ImportError(f'{name} requires the evoxbench package and its database, but got "{e}" when importing')
    return getattr(test_suites, name)(problem_id)


class C10MOP(EvoXBenchProblem):
    def __init__(self, problem_id):
        super().__init__(_suite("c10mop", problem_id, 9))


class CitySegMOP(EvoXBenchProblem):
    def __init__(self, problem_id):
        super().__init__(_suite("citysegmop", problem_id, 15))


class IN1kMOP(EvoXBenchProblem):
    def __init__(self, problem_id):
        super().__init__(_suite("in1kmop", problem_id, 9))
