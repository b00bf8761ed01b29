"""Exact Gaussian-process regression and Laplace-approximated classification on torch tensors.

The reference wraps the ``gpjax`` package (``src/evox/operators/gaussian_process/regression.py:18-107``,
``classification.py:16-102``); here the models are self-contained: kernels with learnable
log-parameters, Cholesky-based conjugate marginal likelihood, hyper-parameters fitted by a
``GradientTransformation`` from ``evox_amd.utils.optimizers`` over autograd gradients."""
from .kernels import RBF, Linear, Matern52, Kernel
from .likelihoods import Gaussian, Bernoulli
from .regression import GPRegression
from .classification import GPClassification

__all__ = ["GPRegression", "GPClassification", "RBF", "Linear", "Matern52", "Kernel", "Gaussian", "Bernoulli"]
