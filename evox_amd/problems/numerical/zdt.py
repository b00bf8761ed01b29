"""ZDT bi-objective suite (Zitzler, Deb & Thiele 2000): ZDT1, 2, 3, 4, 6 with analytic fronts.
Reference: ``src/evox/problems/numerical/zdt.py:7-100``.  Populations are evaluated as whole
(n, d) tensors: f1 = f1(x), f2 = g(x) · h(f1, g)."""
from __future__ import annotations

import math

import torch

from ... import _native
from ...core.problem import Problem
from ...ops._util import check as _check
from ...ops._util import hip as _hip
from ...ops._util import stream_of as _stream_of

_native.register_hip_api({"evox_hip_zdt": (_native.I, [_native.P, _native.I64, _native.I, _native.I, _native.I, _native.P,
                                                      _native.P])})


class ZDTTestSuit(Problem):
    def __init__(self, n, ref_num=100):
        super().__init__()
        self.n = n
        self.ref_num = ref_num

    def _f1(self, x):
        return x[:, 0]

    def _g(self, x):
        return 1 + 9 * x[:, 1:].mean(1)

    def _h(self, f1, g):
        return 1 - torch.sqrt(f1 / g)

    _KIND = 0                                  # fused-kernel selector of the concrete problem (0: none)

    def evaluate(self, state, X):
        if self._KIND and X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1 and X.shape[1] >= 2:
            # one pass over the rows (csrc/moproblems.hip::zdt_kernel) instead of a dozen tensor operations
            n, d = X.shape
            out = torch.empty((n, 2), dtype=torch.float32, device=X.device)
            if n:
                _check(_hip().evox_hip_zdt(X.data_ptr(), X.stride(0), n, d, self._KIND, out.data_ptr(), _stream_of(X)), "zdt")
            return out, state
        f1 = self._f1(X)
        g = self._g(X)
        return torch.stack([f1, g * self._h(f1, g)], dim=1), state

    def pf(self):
        x = torch.linspace(0, 1, self.ref_num)
        return torch.stack([x, 1 - torch.sqrt(x)], dim=1)


class ZDT1(ZDTTestSuit):
    _KIND = 1


class ZDT2(ZDTTestSuit):
    _KIND = 2

    def _h(self, f1, g):
        return 1 - (f1 / g) ** 2

    def pf(self):
        x = torch.linspace(0, 1, self.ref_num)
        return torch.stack([x, 1 - x ** 2], dim=1)


class ZDT3(ZDTTestSuit):
    _KIND = 3

    def _h(self, f1, g):
        return 1 - torch.sqrt(f1 / g) - (f1 / g) * torch.sin(10 * math.pi * f1)

    def pf(self):
        regions = torch.tensor([[0, 0.0830], [0.1822, 0.2577], [0.4093, 0.4538], [0.6183, 0.6525], [0.8233, 0.8518]])
        k = int(self.ref_num / len(regions))
        t = torch.linspace(0, 1, k)[:, None]
        f1 = (regions[:, 0] + t * (regions[:, 1] - regions[:, 0])).T.reshape(-1)
        f2 = 1 - torch.sqrt(f1) - f1 * torch.sin(10 * math.pi * f1)
        return torch.stack([f1, f2], dim=1)


class ZDT4(ZDTTestSuit):
    _KIND = 4

    def _g(self, x):
        r = x[:, 1:]
        return 1 + 10 * (self.n - 1) + (r ** 2 - 10 * torch.cos(4 * math.pi * r)).sum(1)


class ZDT6(ZDTTestSuit):
    _KIND = 6

    def _f1(self, x):
        return 1 - torch.exp(-4 * x[:, 0]) * torch.sin(6 * math.pi * x[:, 0]) ** 6

    def _g(self, x):
        return 1 + 9 * (x[:, 1:].sum(1) / 9) ** 0.25

    def _h(self, f1, g):
        return 1 - (f1 / g) ** 2

    def pf(self):
        f1 = torch.linspace(0.280775, 1, self.ref_num)
        return torch.stack([f1, 1 - f1 ** 2], dim=1)
