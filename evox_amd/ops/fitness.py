"""Single-objective benchmark functions as one fused row-reduction kernel (HIP on GPU).

Each function maps an ``(N, D)`` population to ``(N,)`` fitness.  The ``*_reference``
functions are the plain PyTorch fp32 definitions used on the host and as the numerics oracle
of the kernels.  Formulas: reference ``src/evox/problems/numerical/{sphere,ackley,rastrigin,
rosenbrock,griewank,schwefel}.py``.
"""
from __future__ import annotations

import math

import torch

from .. import _native
from ._util import DTYPE_CODE, check, hip, on_gpu, stream_of

_native.register_hip_api({
    "evox_hip_so_fitness": (_native.I, [_native.I, _native.I, _native.P, _native.P, _native.I64, _native.I,
                                         _native.I64, _native.F, _native.F, _native.F, _native.P]),
})

SPHERE, ACKLEY, RASTRIGIN, ROSENBROCK, GRIEWANK, SCHWEFEL = range(6)


def _launch(func: int, x: torch.Tensor, p0=0.0, p1=0.0, p2=0.0) -> torch.Tensor:
    if x.dim() != 2:
        raise ValueError(f"expected an (N, D) population, got shape {tuple(x.shape)}")
    if x.dtype not in DTYPE_CODE:
        x = x.to(torch.float32)
    if x.stride(1) != 1:
        x = x.contiguous()
    n, d = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    check(hip().evox_hip_so_fitness(func, DTYPE_CODE[x.dtype], x.data_ptr(), out.data_ptr(), n, d, x.stride(0),
                                    float(p0), float(p1), float(p2), stream_of(x)), "so_fitness")
    return out


def sphere_reference(x):
    x = x.float()
    return (x * x).sum(1)


def ackley_reference(x, a=20.0, b=0.2, c=2 * math.pi):
    x = x.float()
    return (-a * torch.exp(-b * torch.sqrt((x * x).mean(1))) - torch.exp(torch.cos(c * x).mean(1)) + a + math.e)


def rastrigin_reference(x):
    x = x.float()
    return 10.0 * x.shape[1] + (x * x - 10.0 * torch.cos(2 * math.pi * x)).sum(1)


def rosenbrock_reference(x):
    x = x.float()
    return (100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (x[:, :-1] - 1.0) ** 2).sum(1)


def griewank_reference(x):
    x = x.float()
    i = torch.arange(1, x.shape[1] + 1, device=x.device, dtype=torch.float32)
    return (x * x).sum(1) / 4000.0 - torch.cos(x / torch.sqrt(i)).prod(1) + 1.0


def schwefel_reference(x):
    x = x.float()
    return 418.9828872724338 * x.shape[1] - (x * torch.sin(torch.sqrt(x.abs()))).sum(1)


def sphere(x):
    return _launch(SPHERE, x) if on_gpu(x) else sphere_reference(x)


def ackley(x, a=20.0, b=0.2, c=2 * math.pi):
    # the kernel compares c against float32(2*pi) to pick the exact cos(pi*t) path
    if on_gpu(x):
        c32 = 6.283185307179586 if abs(c - 2 * math.pi) < 1e-12 else c
        return _launch(ACKLEY, x, a, b, c32)
    return ackley_reference(x, a, b, c)


def rastrigin(x):
    return _launch(RASTRIGIN, x) if on_gpu(x) else rastrigin_reference(x)


def rosenbrock(x):
    return _launch(ROSENBROCK, x) if on_gpu(x) else rosenbrock_reference(x)


def griewank(x):
    return _launch(GRIEWANK, x) if on_gpu(x) else griewank_reference(x)


def schwefel(x):
    return _launch(SCHWEFEL, x) if on_gpu(x) else schwefel_reference(x)
