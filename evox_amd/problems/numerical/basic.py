"""Classic single-objective test functions (reference: ``src/evox/problems/numerical/
{ackley,sphere,rastrigin,rosenbrock,griewank,schwefel}.py``).  Evaluation is one fused
row-reduction kernel per population (``evox_amd.ops.fitness``)."""
from __future__ import annotations

import math

from ...core.problem import Problem
from ...ops import fitness as _f


def ackley_func(a, b, c, x):
    return _f.ackley(x, a, b, c)


def sphere_func(x):
    return _f.sphere(x)


def rastrigin_func(x):
    return _f.rastrigin(x)


def rosenbrock_func(x):
    return _f.rosenbrock(x)


def griewank_func(x):
    return _f.griewank(x)


def schwefel_func(x):
    return _f.schwefel(x)


class Ackley(Problem):
    """``-a exp(-b sqrt(mean x²)) - exp(mean cos(c x)) + a + e``; minimum 0 at the origin."""

    def __init__(self, a: float = 20.0, b: float = 0.2, c: float = 2 * math.pi):
        super().__init__()
        self.a, self.b, self.c = a, b, c

    def evaluate(self, state, X):
        return _f.ackley(X, self.a, self.b, self.c), state


class Sphere(Problem):
    def evaluate(self, state, X):
        return _f.sphere(X), state


class Rastrigin(Problem):
    def evaluate(self, state, X):
        return _f.rastrigin(X), state


class Rosenbrock(Problem):
    def evaluate(self, state, X):
        return _f.rosenbrock(X), state


class Griewank(Problem):
    def evaluate(self, state, X):
        return _f.griewank(X), state


class Schwefel(Problem):
    def evaluate(self, state, X):
        return _f.schwefel(X), state
