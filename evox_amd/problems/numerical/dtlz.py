"""DTLZ1–7 scalable many-objective suite (Deb et al. 2005).  Defaults follow the reference
(``src/evox/problems/numerical/dtlz.py:7-352``): m = 3 objectives, d = m+4 (DTLZ1), m+9
(DTLZ2–6), m+19 (DTLZ7).  Each problem is a distance function g over the last d−m+1 variables
times a position function over the first m−1."""
from __future__ import annotations

import math

import torch

from ... import _native
from ...core.problem import Problem
from ...core.state import State
from ...operators.sampling import GridSampling, UniformSampling
from ...ops._util import check, hip, on_gpu, stream_of

_native.register_hip_api({"evox_hip_dtlz": (_native.I, [_native.P, _native.I64, _native.I, _native.I, _native.I, _native.I,
                                                       _native.P, _native.P])})
_KIND = {}        # class -> kernel variant; filled below (subclasses that override pieces use the tensor path)


def _fused(problem, X):
    """Objectives from the one-pass kernel (``csrc/moproblems.hip``) when ``problem`` is one of the
    seven stock classes and ``X`` is an fp32 device matrix; None otherwise."""
    kind = _KIND.get(type(problem))
    if kind is None or not on_gpu(X) or X.dtype != torch.float32 or X.dim() != 2 or problem.m > 16 \
            or X.shape[1] < problem.m or X.shape[0] == 0:
        return None
    if X.stride(1) != 1:
        X = X.contiguous()
    out = torch.empty((X.shape[0], problem.m), dtype=torch.float32, device=X.device)
    check(hip().evox_hip_dtlz(X.data_ptr(), X.stride(0), X.shape[0], X.shape[1], problem.m, kind, out.data_ptr(),
                              stream_of(X)), "dtlz")
    return out


def _chain(head, tail):
    """``f_j = Π_{i < m−1−j} head_i · tail_j`` — the shared triangular product of the suite.
    head: (n, m−1) factors accumulated left to right, tail: (n, m−1) factor applied last
    (column j of the result uses tail column m−1−j; column 0 uses none)."""
    n = head.shape[0]
    one = torch.ones((n, 1), dtype=head.dtype, device=head.device)
    prod = torch.cumprod(torch.cat([one, head], dim=1), dim=1).flip(1)          # (n, m)
    last = torch.cat([one, tail.flip(1)], dim=1)
    return prod * last


class DTLZTestSuit(Problem):
    default_extra = 9

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__()
        self.m = 3 if m is None else m
        self.d = self.m + self.default_extra if d is None else d
        self.ref_num = ref_num
        self.sample = UniformSampling(self.ref_num * self.m, self.m)

    def setup(self, key):
        return State(key=key)

    def _g(self, xm):
        return ((xm - 0.5) ** 2).sum(1, keepdim=True)

    def _position(self, X):
        return X[:, : self.m - 1]

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = self._g(X[:, m - 1:])
        pos = self._position(X) if not hasattr(self, "_position_with_g") else self._position_with_g(X, g)
        ang = pos * (math.pi / 2)
        f = (1 + g) * _chain(torch.cos(ang).clamp_min(0), torch.sin(ang))
        return f, state

    def pf(self):
        f = self.sample()[0]
        return f / torch.linalg.vector_norm(f, dim=1, keepdim=True)


def _multimodal_g(xm):
    k = xm.shape[1]
    return 100 * (k + ((xm - 0.5) ** 2 - torch.cos(20 * math.pi * (xm - 0.5))).sum(1, keepdim=True))


class DTLZ1(DTLZTestSuit):
    default_extra = 4

    def __init__(self, d=None, m=None, ref_num=100):
        super().__init__(d, m, ref_num)

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = _multimodal_g(X[:, m - 1:])
        pos = X[:, : m - 1]
        return 0.5 * (1 + g) * _chain(pos, 1 - pos), state

    def pf(self):
        return self.sample()[0] / 2


class DTLZ2(DTLZTestSuit):
    pass


class DTLZ3(DTLZ2):
    def _g(self, xm):
        return _multimodal_g(xm)


class DTLZ4(DTLZ2):
    def _position(self, X):
        return X[:, : self.m - 1] ** 100


class DTLZ5(DTLZTestSuit):
    def _position_with_g(self, X, g):
        m = self.m
        first = X[:, :1]
        rest = (1 + 2 * g * X[:, 1: m - 1]) / (2 + 2 * g)
        return torch.cat([first, rest], dim=1)

    def pf(self):
        n = self.ref_num * self.m
        t = torch.linspace(0, 1, n)
        f = torch.stack([t, 1 - t], dim=1)
        f = f / torch.linalg.vector_norm(f, dim=1, keepdim=True)
        f = torch.cat([f[:, :1].expand(-1, self.m - 2), f], dim=1)
        expo = torch.cat([torch.tensor([self.m - 2.0]), torch.arange(self.m - 2, -1, -1.0)])
        return f / math.sqrt(2) ** expo


class DTLZ6(DTLZ5):
    def _g(self, xm):
        return (xm ** 0.1).sum(1, keepdim=True)


class DTLZ7(DTLZTestSuit):
    default_extra = 19

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(d, m, ref_num)
        self.sample = GridSampling(self.ref_num * self.m, self.m - 1)

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = 1 + 9 * X[:, m - 1:].mean(1, keepdim=True)
        head = X[:, : m - 1]
        last = (1 + g) * (m - (head / (1 + g) * (1 + torch.sin(3 * math.pi * head))).sum(1, keepdim=True))
        return torch.cat([head, last], dim=1), state

    def pf(self):
        a, b, c, d = 0.0, 0.251412, 0.631627, 0.859401
        median = (b - a) / (d - c + b - a)
        x = self.sample()[0]
        x = torch.where(x <= median, x * (b - a) / median + a, (x - median) * (d - c) / (1 - median) + c)
        last = 2 * (self.m - (x / 2 * (1 + torch.sin(3 * math.pi * x))).sum(1, keepdim=True))
        return torch.cat([x, last], dim=1)


_KIND.update({DTLZ1: 1, DTLZ2: 2, DTLZ3: 3, DTLZ4: 4, DTLZ5: 5, DTLZ6: 6, DTLZ7: 7})
