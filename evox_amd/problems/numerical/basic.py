"""Classic single-objective test functions (reference: ``src/evox/problems/numerical/
{ackley,sphere,rastrigin,rosenbrock,griewank,schwefel}.py``).  Evaluation is one fused
row-reduction kernel per population (``evox_amd.ops.fitness``)."""
from __future__ import annotations

import math

import torch

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


class Ellipsoid(Problem):
    """``Σ_i cond^{i/(d-1)} x_i²`` — the axis-parallel ill-conditioned quadratic (condition number ``cond``, 10⁶ by
    default) that covariance-adapting strategies are tuned on.  Not part of the reference's numerical suite; it is the
    stress case for the factorisation of an ill-conditioned covariance in ``bench.py --problem ellipsoid``."""

    def __init__(self, cond: float = 1e6):
        super().__init__()
        self.cond = float(cond)
        self._w = None

    def evaluate(self, state, X):
        d = X.shape[-1]
        if self._w is None or self._w.shape[0] != d or self._w.device != X.device:
            e = torch.arange(d, dtype=torch.float64) / max(d - 1, 1)
            self._w = torch.pow(torch.tensor(self.cond, dtype=torch.float64), e).to(torch.float32).to(X.device)
        Xf = X.float()
        return (Xf * Xf) @ self._w, state
