from __future__ import annotations

import math

import torch

from ...core.random import PRNGKey
from ...utils import optimizers as optim
from .kernels import RBF
from .likelihoods import Gaussian


def _fit_params(params, loss_fn, optimizer, num_iters):
    """Minimise ``loss_fn(params)`` over a dict of tensors with a GradientTransformation."""
    names = sorted(params)
    leaves = [params[k].detach().clone().requires_grad_(True) for k in names]
    opt_state = optimizer.init(leaves)
    history = []
    for _ in range(num_iters):
        loss = loss_fn(dict(zip(names, leaves)))
        grads = torch.autograd.grad(loss, leaves)
        updates, opt_state = optimizer.update(list(grads), opt_state, leaves)
        leaves = [(p + u).detach().requires_grad_(True) for p, u in zip(leaves, updates)]
        history.append(float(loss.detach()))
    return {k: v.detach() for k, v in zip(names, leaves)}, history


class GPRegression:
    """Zero-mean GP regression, hyper-parameters fitted by minimising the negative conjugate
    marginal log-likelihood for 250 optimiser steps (``regression.py:64-91`` in the reference).
    ``predict`` returns ``((mean, covariance), mean, stddev)`` of the *predictive* distribution
    (latent variance plus observation noise)."""

    def __init__(self, kernel=None, mean_fun=None, likelihood=None, object=None, key=None, num_iters=250):
        self.kernel = kernel if kernel is not None else RBF()
        self.mean_fun = mean_fun
        self.likelihood = likelihood if likelihood is not None else Gaussian()
        self.object = object
        self.key = key if key is not None else PRNGKey(123)
        self.num_iters = num_iters

    def _nmll(self, params, x, y):
        n = x.shape[0]
        K = self.kernel(params, x, x) + (params["obs_noise"].exp() ** 2 + 1e-6) * torch.eye(n, device=x.device)
        L = torch.linalg.cholesky(K)
        alpha = torch.cholesky_solve(y, L)
        return (0.5 * (y * alpha).sum() + torch.log(torch.diagonal(L)).sum() * y.shape[1]
                + 0.5 * n * y.shape[1] * math.log(2 * math.pi))

    def fit(self, x, y, optimzer=None):
        x = torch.as_tensor(x, dtype=torch.float32)
        y = torch.as_tensor(y, dtype=torch.float32).to(x.device)
        if y.dim() == 1:
            y = y[:, None]
        self.x, self.y = x, y
        params = self.kernel.init_params(x.device)
        params["obs_noise"] = torch.tensor(math.log(self.likelihood.obs_stddev), device=x.device)
        optimizer = optimzer if optimzer is not None else optim.sgd(0.001)
        self.params, self.history = _fit_params(params, lambda p: self._nmll(p, x, y), optimizer, self.num_iters)
        n = x.shape[0]
        K = self.kernel(self.params, x, x) + (self.params["obs_noise"].exp() ** 2 + 1e-6) * torch.eye(n, device=x.device)
        self._L = torch.linalg.cholesky(K)
        self._alpha = torch.cholesky_solve(y, self._L)
        return self

    def predict(self, x):
        x = torch.as_tensor(x, dtype=torch.float32).to(self.x.device)
        Ks = self.kernel(self.params, x, self.x)
        mean = Ks @ self._alpha
        v = torch.linalg.solve_triangular(self._L, Ks.T, upper=False)
        cov = self.kernel(self.params, x, x) - v.T @ v
        cov = cov + self.params["obs_noise"].exp() ** 2 * torch.eye(x.shape[0], device=x.device)
        std = torch.diagonal(cov).clamp_min(0).sqrt()
        mean = mean[:, 0] if mean.shape[1] == 1 else mean
        return (mean, cov), mean, std
