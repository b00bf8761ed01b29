"""LSMOP1–9: large-scale multi-/many-objective test problems (Cheng, Jin & Olhofer 2017).
Reference: ``src/evox/problems/numerical/lsmop.py:17-454``.

The decision vector is split into m−1 position variables and m groups of distance variables
(group sizes from a chaotic logistic sequence), each group further split into ``nk`` = 5
sub-components.  A linear (LSMOP1–4) or non-linear (LSMOP5–9) *variable linkage* ties every
distance variable to x₁.  All sub-component sums are evaluated for the whole population with
one pass per group.

The group layout is derived from the actual number of variables ``d`` (default 100·m, the
paper's setting).  The reference derives it from 100·m and then overrides ``d`` with m+4 in
each subclass (``lsmop.py:110-119``), which leaves groups pointing past the end of the vector."""
from __future__ import annotations

import math
from itertools import cycle

import torch

from ... import _native
from ...core.problem import Problem
from ...core.state import State
from ...operators.sampling import UniformSampling
from ...ops._util import check, hip, on_gpu, stream_of
from .dtlz import _chain

_native.register_hip_api({"evox_hip_mo_grouped": (_native.I, [_native.P, _native.I64, _native.I, _native.I, _native.I, _native.I,
                                                             _native.P, _native.P, _native.I, _native.I, _native.I, _native.P,
                                                             _native.P])})


def _sphere(x):
    return (x * x).sum(1)


def _griewank(x):
    i = torch.arange(1, x.shape[1] + 1, device=x.device, dtype=x.dtype)
    return (x * x).sum(1) / 4000 - torch.cos(x / torch.sqrt(i)).prod(1) + 1


def _rosenbrock(x):
    return (100 * (x[:, :-1] ** 2 - x[:, 1:]) ** 2 + (x[:, :-1] - 1) ** 2).sum(1)


def _ackley(x):
    d = max(x.shape[1], 1)
    return (-20 * torch.exp(-0.2 * torch.sqrt((x * x).sum(1) / d)) - torch.exp(torch.cos(2 * math.pi * x).sum(1) / d)
            + 20 + math.e)


def _schwefel(x):
    return x.abs().amax(1)


def _rastrigin(x):
    return (x * x - 10 * torch.cos(2 * math.pi * x) + 10).sum(1)


def chaotic_group_sizes(m, n_vars, nk):
    """Sub-component length of each of the m groups (logistic map a=3.8, c₀=0.1)."""
    c = [3.8 * 0.1 * (1 - 0.1)]
    for _ in range(1, m):
        c.append(3.8 * c[-1] * (1 - c[-1]))
    total = sum(c)
    return [int(math.floor(ci / total * n_vars / nk)) for ci in c]


def grouped_g(x, m, sublen, nk, funcs):
    """(N, m) matrix of group landscape values, each averaged over its ``nk`` sub-components."""
    cols = []
    start = m - 1
    for size, fn in zip(sublen, cycle(funcs)):
        block = x[:, start:start + nk * size].reshape(x.shape[0] * nk, size) if size > 0 else None
        if block is None:
            cols.append(torch.zeros(x.shape[0], device=x.device, dtype=x.dtype))
        else:
            cols.append(fn(block).reshape(x.shape[0], nk).sum(1) / (size * nk))
        start += nk * size
    return torch.stack(cols, dim=1)


_LANDSCAPE_ID = {_sphere: 0, _griewank: 1, _rosenbrock: 2, _ackley: 3, _schwefel: 4, _rastrigin: 5}
FRONT_LINEAR, FRONT_CONCAVE, FRONT_DISCONNECTED, FRONT_INVERTED = 0, 1, 2, 3


def fused_grouped(X, m, nk, sublen, funcs, nonlinear, front):
    """Objectives of a grouped large-scale problem from the one-pass kernel (``csrc/mo_large.hip``: a wavefront per
    individual, linkage, landscapes and front in one launch), or None when ``X`` is not an fp32 device matrix."""
    import ctypes

    if not on_gpu(X) or X.dtype != torch.float32 or X.dim() != 2 or m > 16 or X.shape[0] == 0 or len(funcs) > 2:
        return None
    if X.stride(1) != 1:
        X = X.contiguous()
    out = torch.empty((X.shape[0], m), dtype=torch.float32, device=X.device)
    lens = (ctypes.c_int * m)(*sublen)
    ids = (ctypes.c_int * len(funcs))(*[_LANDSCAPE_ID[f] for f in funcs])
    check(hip().evox_hip_mo_grouped(X.data_ptr(), X.stride(0), X.shape[0], X.shape[1], m, nk, lens, ids, len(funcs),
                                    int(bool(nonlinear)), front, out.data_ptr(), stream_of(X)), "mo_grouped")
    return out


def _dtlz7_front(n, m, dtype=torch.float32):
    lo1, hi1, lo2, hi2 = 0.0, 0.251412, 0.631627, 0.859401
    median = (hi1 - lo1) / (hi2 - lo2 + hi1 - lo1)
    gap = torch.linspace(0, 1, int(math.ceil(n ** (1 / (m - 1)))), dtype=dtype)
    x = torch.stack([g.reshape(-1) for g in torch.meshgrid(*([gap] * (m - 1)), indexing="xy")], dim=1)
    x = torch.where(x <= median, x * (hi1 - lo1) / median + lo1, (x - median) * (hi2 - lo2) / (1 - median) + lo2)
    last = 2 * (m - (x / 2 * (1 + torch.sin(3 * math.pi * x))).sum(1, keepdim=True))
    return torch.cat([x, last], dim=1)


class LSMOP(Problem):
    nk = 5
    funcs = (_sphere,)
    nonlinear_linkage = False

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__()
        self.m = 3 if m is None else m
        self.d = self.m * 100 if d is None else d
        self.ref_num = ref_num
        self.sublen = tuple(chaotic_group_sizes(self.m, self.d - self.m + 1, self.nk))
        bounds = [0]
        for s in self.sublen:
            bounds.append(bounds[-1] + s * self.nk)
        self.len = tuple(bounds)

    def setup(self, key):
        return State(key=key)

    def _link(self, X):
        m, d = self.m, X.shape[1]
        idx = torch.arange(m, d + 1, device=X.device, dtype=X.dtype) / d
        w = 1 + (torch.cos(idx * math.pi / 2) if self.nonlinear_linkage else idx)
        tail = w * X[:, m - 1:] - X[:, :1] * 10
        return torch.cat([X[:, : m - 1], tail], dim=1)

    def _g(self, X):
        return grouped_g(X, self.m, self.sublen, self.nk, self.funcs)

    def _fused(self, X, front):
        # only the stock classes: a subclass may override the linkage or the landscapes
        if type(self) not in _STOCK or X.shape[1] != self.d:
            return None
        return fused_grouped(X, self.m, self.nk, self.sublen, self.funcs, self.nonlinear_linkage, front)

    def evaluate(self, state, X):
        f = self._fused(X, FRONT_CONCAVE if self.nonlinear_linkage else FRONT_LINEAR)
        if f is not None:
            return f, state
        X = self._link(X)
        g = self._g(X)
        pos = X[:, : self.m - 1]
        if self.nonlinear_linkage:
            gg = 1 + g + torch.cat([g[:, 1:], torch.zeros_like(g[:, :1])], dim=1)
            ang = pos * (math.pi / 2)
            return gg * _chain(torch.cos(ang), torch.sin(ang)), state
        return (1 + g) * _chain(pos, 1 - pos), state

    def pf(self):
        """Unit simplex (linear linkage) or unit sphere (non-linear linkage): the image of the
        solutions with all group landscapes at zero.  The reference halves the simplex
        (``lsmop.py:66-68``), which does not match its own objective functions."""
        f = UniformSampling(self.ref_num * self.m, self.m)()[0]
        if self.nonlinear_linkage:
            f = f / torch.linalg.vector_norm(f, dim=1, keepdim=True)
        return f


class LSMOP1(LSMOP):
    funcs = (_sphere,)


class LSMOP2(LSMOP):
    funcs = (_griewank, _schwefel)


class LSMOP3(LSMOP):
    funcs = (_rastrigin, _rosenbrock)


class LSMOP4(LSMOP):
    funcs = (_ackley, _griewank)


class LSMOP5(LSMOP):
    funcs = (_sphere,)
    nonlinear_linkage = True


class LSMOP6(LSMOP):
    funcs = (_rosenbrock, _schwefel)
    nonlinear_linkage = True


class LSMOP7(LSMOP):
    funcs = (_ackley, _rosenbrock)
    nonlinear_linkage = True


class LSMOP8(LSMOP):
    funcs = (_griewank, _sphere)
    nonlinear_linkage = True


class LSMOP9(LSMOP):
    """Disconnected front (DTLZ7 shape) over a sphere/Ackley landscape."""
    funcs = (_sphere, _ackley)
    nonlinear_linkage = True

    def evaluate(self, state, X):
        f = self._fused(X, FRONT_DISCONNECTED)
        if f is not None:
            return f, state
        m = self.m
        X = self._link(X)
        g = 1 + self._g(X).sum(1, keepdim=True)
        head = X[:, : m - 1]
        last = (1 + g) * (m - (head / (1 + g) * (1 + torch.sin(3 * math.pi * head))).sum(1, keepdim=True))
        return torch.cat([head, last], dim=1), state

    def pf(self):
        return _dtlz7_front(self.ref_num * self.m, self.m)


_STOCK = {LSMOP1, LSMOP2, LSMOP3, LSMOP4, LSMOP5, LSMOP6, LSMOP7, LSMOP8, LSMOP9}
