"""Covariance functions with positive parameters stored as logs."""
from __future__ import annotations

import math

import torch


class Kernel:
    param_names: tuple = ()

    def init_params(self, device=None):
        return {k: torch.tensor(math.log(v), dtype=torch.float32, device=device) for k, v in self._init.items()}

    def __call__(self, params, x, y):
        raise NotImplementedError


class RBF(Kernel):
    def __init__(self, lengthscale=1.0, variance=1.0):
        self._init = {"lengthscale": float(lengthscale), "variance": float(variance)}

    def __call__(self, params, x, y):
        ls, var = params["lengthscale"].exp(), params["variance"].exp()
        d2 = torch.cdist(x / ls, y / ls).square()
        return var * torch.exp(-0.5 * d2)


class Matern52(Kernel):
    def __init__(self, lengthscale=1.0, variance=1.0):
        self._init = {"lengthscale": float(lengthscale), "variance": float(variance)}

    def __call__(self, params, x, y):
        ls, var = params["lengthscale"].exp(), params["variance"].exp()
        r = math.sqrt(5.0) * torch.cdist(x / ls, y / ls)
        return var * (1 + r + r.square() / 3) * torch.exp(-r)


class Linear(Kernel):
    def __init__(self, variance=1.0):
        self._init = {"variance": float(variance)}

    def __call__(self, params, x, y):
        return params["variance"].exp() * (x @ y.T)
