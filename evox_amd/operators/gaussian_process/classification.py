from __future__ import annotations

import torch

from ...core.random import PRNGKey
from ...utils import optimizers as optim
from .kernels import RBF
from .likelihoods import Bernoulli
from .regression import _fit_params


class GPClassification:
    """Binary GP classifier.  The latent values at the training inputs are whitened
    (``f = L·v``) and fitted jointly with the kernel parameters by minimising the negative log
    posterior density for 500 optimiser steps, the objective the reference selects
    (``classification.py:48-49,71-81``); predictions use the MAP latent with the probit-style
    variance correction of the logistic link."""

    def __init__(self, kernel=None, mean_fun=None, likelihood=None, object=None, key=None, num_iters=500):
        self.kernel = kernel if kernel is not None else RBF()
        self.mean_fun = mean_fun
        self.likelihood = likelihood if likelihood is not None else Bernoulli()
        self.object = object
        self.key = key if key is not None else PRNGKey(123)
        self.num_iters = num_iters

    def _chol(self, params, x):
        K = self.kernel(params, x, x) + 1e-4 * torch.eye(x.shape[0], device=x.device)
        return torch.linalg.cholesky(K)

    def _neg_log_posterior(self, params, x, y):
        kp = {k: v for k, v in params.items() if k != "whitened"}
        f = self._chol(kp, x) @ params["whitened"]
        log_lik = -(torch.nn.functional.softplus(-f) * y + torch.nn.functional.softplus(f) * (1 - y)).sum()
        log_prior = -0.5 * params["whitened"].square().sum()
        return -(log_lik + log_prior)

    def fit(self, x, y, optimizer=None):
        x = torch.as_tensor(x, dtype=torch.float32)
        y = torch.as_tensor(y, dtype=torch.float32).to(x.device).reshape(-1, 1)
        self.x, self.y = x, y
        params = self.kernel.init_params(x.device)
        params["whitened"] = torch.zeros((x.shape[0], 1), device=x.device)
        opt = optimizer if optimizer is not None else optim.adam(0.01)
        if callable(opt) and not hasattr(opt, "init"):
            opt = opt(0.01)
        self.params, self.history = _fit_params(params, lambda p: self._neg_log_posterior(p, x, y), opt, self.num_iters)
        self._kp = {k: v for k, v in self.params.items() if k != "whitened"}
        self._L = self._chol(self._kp, x)
        return self

    def predict(self, x):
        x = torch.as_tensor(x, dtype=torch.float32).to(self.x.device)
        Ks = self.kernel(self._kp, x, self.x)
        v = torch.linalg.solve_triangular(self._L, Ks.T, upper=False)                 # L^-1 k*
        latent_mean = (v.T @ self.params["whitened"])[:, 0]
        # Laplace curvature of the logistic likelihood at the MAP latent
        f = (self._L @ self.params["whitened"])[:, 0]
        w = torch.sigmoid(f) * (1 - torch.sigmoid(f))
        n = self.x.shape[0]
        K = self._L @ self._L.T
        B = torch.eye(n, device=x.device) + w.sqrt()[:, None] * K * w.sqrt()[None, :]
        LB = torch.linalg.cholesky(B)
        u = torch.linalg.solve_triangular(LB, w.sqrt()[:, None] * Ks.T, upper=False)
        latent_var = (torch.diagonal(self.kernel(self._kp, x, x)) - u.square().sum(0)).clamp_min(0)
        p = torch.sigmoid(latent_mean / torch.sqrt(1 + torch.pi * latent_var / 8))
        std = torch.sqrt(p * (1 - p))
        return (latent_mean, latent_var), p, std
