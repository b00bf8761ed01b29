"""CEC 2022 single-objective bound-constrained test suite, F1–F12, for D ∈ {2, 10, 20}
(hybrid functions F6–F8: D ∈ {10, 20}).
Reference: ``src/evox/problems/numerical/cec2022_so.py:10-766``.

Every function is evaluated for the whole population at once: shift/scale/rotate is one small
GEMM ``((X − o)·s) Mᵀ`` (on the GPU a single kernel with the matrix in LDS, ``csrc/moproblems.hip``), the basic functions are row-wise reductions over (N, D) tensors.  The
benchmark constants (shift vectors, rotation matrices, shuffles) are the official ones, packed
in ``data/cec2022.npz`` (see ``tools/pack_cec2022.py``).

Two departures from the reference, both towards the official definition: the composition
weights fall back to a one-hot weight when a point coincides with a component optimum (the
reference produces NaN there), and F12's sixth component uses its own shift/rotation block
(the reference re-uses the fifth, ``cec2022_so.py:724-725``)."""
from __future__ import annotations

import math
from functools import lru_cache
from pathlib import Path

import numpy as np
import torch

from ... import _native
from ...core.problem import Problem
from ...ops._util import check as _check
from ...ops._util import hip as _hip
from ...ops._util import stream_of as _stream_of

_native.register_hip_api({"evox_hip_shift_rotate": (_native.I, [_native.P, _native.I64, _native.P, _native.P, _native.P,
                                                               _native.F, _native.I, _native.I, _native.P, _native.P])})

_native.register_hip_api({"evox_hip_cec_simple": (_native.I, [_native.P, _native.I64, _native.P, _native.P, _native.F, _native.I,
                                                             _native.F, _native.I, _native.I, _native.P, _native.P])})

_native.register_hip_api({"evox_hip_cec_full": (_native.I, [_native.P, _native.I64, _native.P, _native.I64, _native.P, _native.I,
                                                           _native.P, _native.I, _native.F, _native.I, _native.I, _native.P,
                                                           _native.P])})

_DATA = Path(__file__).parent / "data" / "cec2022.npz"


@lru_cache(maxsize=1)
def _archive():
    with np.load(_DATA) as z:
        return {k: z[k] for k in z.files}


# ------------------------------------------------------------------------- basic functions
def zakharov(z):
    i = torch.arange(1, z.shape[1] + 1, device=z.device, dtype=z.dtype)
    t1 = (z * z).sum(1)
    t2 = (0.5 * i * z).sum(1)
    return t1 + t2 ** 2 + t2 ** 4


def rosenbrock(z):
    z = z + 1
    return 100 * ((z[:, :-1] ** 2 - z[:, 1:]) ** 2).sum(1) + ((1 - z[:, :-1]) ** 2).sum(1)


def schaffer_f7(y):
    nx = y.shape[1]
    s = torch.sqrt(y[:, :-1] ** 2 + y[:, 1:] ** 2)
    f = (torch.sqrt(s) * (1 + torch.sin(50.0 * s ** 0.2) ** 2)).sum(1)
    return f * f / (nx - 1) / (nx - 1)


def rastrigin(z):
    z = z * 0.0512
    return (z * z - 10 * torch.cos(2 * math.pi * z) + 10).sum(1)


def levy(z):
    w = 1 + z / 4
    w1, wl = w[:, :-1], w[:, -1]
    return (torch.sin(math.pi * w[:, 0]) ** 2 + (((w1 - 1) ** 2) * (1 + 10 * torch.sin(math.pi * w1 + 1) ** 2)).sum(1)
            + ((wl - 1) ** 2) * (1 + torch.sin(2 * math.pi * wl) ** 2))


def bent_cigar(z):
    return z[:, 0] ** 2 + 1e6 * (z[:, 1:] ** 2).sum(1)


def hgbat(z):
    d = z.shape[1]
    z = z * 0.05 - 1
    sq, s = (z * z).sum(1), z.sum(1)
    return torch.sqrt((sq ** 2 - s ** 2).abs()) + (0.5 * sq + s) / d + 0.5


def katsuura(z):
    nx = z.shape[1]
    z = z * 0.05
    p = 2.0 ** torch.arange(1, 33, device=z.device, dtype=z.dtype)                     # (32,)
    t = p * z[:, :, None]
    temp = ((t - torch.floor(t + 0.5)).abs() / p).sum(2)                               # (N, nx)
    i = torch.arange(1, nx + 1, device=z.device, dtype=z.dtype)
    f = ((1 + i * temp) ** (10.0 / nx ** 1.2)).prod(1)
    c = 10.0 / nx / nx
    return f * c - c


def ackley(z):
    d = z.shape[1]
    return (-20 * torch.exp(-0.2 * torch.sqrt((z * z).sum(1) / d)) - torch.exp(torch.cos(2 * math.pi * z).sum(1) / d)
            + 20 + math.e)


def schwefel(z):
    nx = z.shape[1]
    z = z * 10 + 4.209687462275036e002
    m_hi = torch.fmod(z, 500)
    m_lo = torch.fmod(z.abs(), 500)
    mid = -z * torch.sin(torch.sqrt(z.abs()))
    hi = -(500.0 - m_hi) * torch.sin(torch.sqrt((500.0 - m_hi).abs())) + ((z - 500.0) / 100) ** 2 / nx
    lo = -(-500.0 + m_lo) * torch.sin(torch.sqrt((500.0 - m_lo).abs())) + ((z + 500.0) / 100) ** 2 / nx
    per = torch.where(z > 500, hi, torch.where(z < -500, lo, mid))
    return per.sum(1) + 4.189828872724338e002 * nx


def happycat(z):
    d = z.shape[1]
    z = z * 0.05 - 1
    sq, s = (z * z).sum(1), z.sum(1)
    return (sq - d).abs() ** 0.25 + (0.5 * sq + s) / d + 0.5


def elliptic(z):
    d = z.shape[1]
    e = 10 ** (6 * torch.arange(d, device=z.device, dtype=z.dtype) / max(d - 1, 1))
    return (e * z * z).sum(1)


def discus(z):
    return 1e6 * z[:, 0] ** 2 + (z[:, 1:] ** 2).sum(1)


def exp_schaffer(z):
    sq = z ** 2 + torch.roll(z, 1, dims=1) ** 2
    return (0.5 + (torch.sin(torch.sqrt(sq)) ** 2 - 0.5) / (1 + 0.001 * sq) ** 2).sum(1)


def exp_grie_rosen(z):
    z = z * 0.05 + 1
    nxt = torch.roll(z, -1, dims=1)
    t = 100.0 * (z * z - nxt) ** 2 + (z - 1.0) ** 2
    return (t * t / 4000.0 - torch.cos(t) + 1.0).sum(1)


def griewank(z):
    i = torch.arange(1, z.shape[1] + 1, device=z.device, dtype=z.dtype)
    return (z * z).sum(1) / 4000 - torch.cos(z / torch.sqrt(i)).prod(1) + 1


# --------------------------------------------------------------------------------- suite
class CEC2022Function(Problem):
    """Base: holds the constants of function ``func_num`` per dimension, moved lazily to the
    device / dtype of the evaluated population."""

    func_num = 0
    parts = None          # hybrid functions: share of the variables per component
    floor = 1e-8          # values below are reported as exactly 0

    def __init__(self):
        super().__init__()
        data = _archive()
        f = self.func_num
        self._shift = data[f"shift_{f}"]
        self._rot = {d: data[f"rot_{f}_{d}"] for d in (2, 10, 20)}
        self._shuffle = {d: data[f"shuffle_{f}_{d}"] - 1 for d in (10, 20) if f"shuffle_{f}_{d}" in data}
        self._cache = {}

    def _consts(self, d, like):
        key = (d, like.device, like.dtype)
        if key not in self._cache:
            if d not in self._rot:
                raise ValueError(f"CEC2022 F{self.func_num} is defined for D in {sorted(self._rot)}, got {d}")
            k = self._rot[d].shape[0] // d
            c = {"o": torch.as_tensor(self._shift[:, :d], device=like.device).to(like.dtype),
                 "M": torch.as_tensor(self._rot[d], device=like.device).to(like.dtype).reshape(k, d, d)}
            if self._shuffle:
                if d not in self._shuffle:
                    raise ValueError(f"CEC2022 F{self.func_num} (hybrid) is defined for D in {sorted(self._shuffle)}")
                c["S"] = torch.as_tensor(self._shuffle[d], device=like.device)
                sizes = [int(round(p * d)) for p in self.parts]
                bounds = np.cumsum([0] + sizes)
                c["groups"] = [slice(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:])]
            self._cache[key] = c
        return self._cache[key]

    @staticmethod
    def _ssr(x, o, M, rate=1.0, shuffle=None):
        """Shift, scale, rotate (and permute): ``z = ((x − o)·rate) Mᵀ``, then ``z[:, shuffle]``."""
        d = x.shape[1]
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and d <= 128 and x.shape[0] > 0 \
                and M.dtype == torch.float32 and (shuffle is None or shuffle.dtype == torch.int64):
            if x.stride(1) != 1:
                x = x.contiguous()
            out = torch.empty((x.shape[0], d), dtype=torch.float32, device=x.device)
            _check(_hip().evox_hip_shift_rotate(x.data_ptr(), x.stride(0), o.contiguous().data_ptr(),
                                                M.contiguous().data_ptr(),
                                                shuffle.contiguous().data_ptr() if shuffle is not None else None,
                                                float(rate), x.shape[0], d, out.data_ptr(), _stream_of(x)), "shift_rotate")
            return out
        z = ((x - o) * rate) @ M.T
        return z if shuffle is None else z[:, shuffle]

    def _f(self, x, c):
        raise NotImplementedError

    def _floor(self, d: int) -> float:
        return self.floor

    def _fused(self, X):
        """Functions 3 and 6-12 in one launch (``csrc/cec2022.hip``): shift, rotations, shuffle, basic functions or
        composition, zero floor.  None when the input is not an fp32 GPU matrix of a dimension the suite defines."""
        f = self.func_num
        if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.shape[0] > 0 and (f == 3 or 6 <= f <= 12)):
            return None
        d = X.shape[1]
        if d not in self._rot or (6 <= f <= 8 and d not in self._shuffle):
            return None
        c = self._consts(d, X)
        if X.stride(1) != 1:
            X = X.contiguous()
        o, M = c["o"].contiguous(), c["M"][:6].contiguous()        # the data files hold ten matrices; at most six are used
        S = c["S"].contiguous() if "S" in c else None
        out = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
        _check(_hip().evox_hip_cec_full(X.data_ptr(), X.stride(0), o.data_ptr(), o.stride(0), M.data_ptr(), int(M.shape[0]),
                                        S.data_ptr() if S is not None else None, f, float(self._floor(d)), X.shape[0], d,
                                        out.data_ptr(), _stream_of(X)), "cec_full")
        return out

    def evaluate(self, state, X):
        X = X.to(torch.float32) if not X.is_floating_point() else X
        import os
        fused = self._fused(X) if os.environ.get("EVOX_AMD_CEC_FUSED", "1") == "1" else None
        if fused is not None:
            return fused, state
        f = self._f(X, self._consts(X.shape[1], X))
        floor = self._floor(X.shape[1])
        return torch.where(f < floor, torch.zeros_like(f), f), state

    def _compose(self, x, fs, c, bias, sigma, lamb):
        d = x.shape[1]
        n = fs.shape[1]
        o = c["o"][:n]
        d2 = ((x[:, None, :] - o[None, :, :]) ** 2).sum(2)                               # (N, n)
        sigma = torch.as_tensor(sigma, device=x.device, dtype=x.dtype)
        w = torch.exp(-0.5 * d2 / (sigma ** 2 * d)) / torch.sqrt(d2)
        at_opt = d2 == 0
        any_opt = at_opt.any(1, keepdim=True)
        w = torch.where(any_opt, at_opt.to(x.dtype), w)
        w = w / w.sum(1, keepdim=True)
        bias = torch.as_tensor(bias, device=x.device, dtype=x.dtype)
        lamb = torch.as_tensor(lamb, device=x.device, dtype=x.dtype)
        return (w * (lamb * fs + bias)).sum(1)


def _simple(num, basic, rate=1.0):
    class _F(CEC2022Function):
        func_num = num

        def _f(self, x, c):
            return basic(self._ssr(x, c["o"][0], c["M"][0], rate))

        def evaluate(self, state, X):
            if X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.shape[1] in (2, 10, 20) and X.shape[0] > 0:
                # shift, rotate, basic function and the zero floor in one launch (csrc/moproblems.hip::cec_simple_kernel)
                c = self._consts(X.shape[1], X)
                if X.stride(1) != 1:
                    X = X.contiguous()
                out = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
                _check(_hip().evox_hip_cec_simple(X.data_ptr(), X.stride(0), c["o"][0].contiguous().data_ptr(),
                                                  c["M"][0].contiguous().data_ptr(), float(rate), num, float(self.floor),
                                                  X.shape[0], X.shape[1], out.data_ptr(), _stream_of(X)), "cec_simple")
                return out, state
            return super().evaluate(state, X)
    _F.__name__ = _F.__qualname__ = f"F{num}_CEC2022"
    return _F


F1_CEC2022 = _simple(1, zakharov)
F2_CEC2022 = _simple(2, rosenbrock, 2.048 / 100.0)
F4_CEC2022 = _simple(4, rastrigin)
F5_CEC2022 = _simple(5, levy)


class F3_CEC2022(CEC2022Function):
    """Expanded Schaffer F7 – evaluated, as in the official code, on the shifted but unrotated
    point."""
    func_num = 3

    def _f(self, x, c):
        return schaffer_f7(x - c["o"][0])


class F6_CEC2022(CEC2022Function):
    func_num = 6
    parts = (0.4, 0.4, 0.2)

    def _f(self, x, c):
        z = self._ssr(x, c["o"][0], c["M"][0], 1.0, c["S"])
        g = c["groups"]
        return bent_cigar(z[:, g[0]]) + hgbat(z[:, g[1]]) + rastrigin(z[:, g[2]])


class F7_CEC2022(CEC2022Function):
    func_num = 7
    parts = (0.1, 0.2, 0.2, 0.2, 0.1, 0.2)

    def _f(self, x, c):
        z = self._ssr(x, c["o"][0], c["M"][0], 1.0, c["S"])
        g = c["groups"]
        last = g[5].stop - g[5].start
        return (hgbat(z[:, g[0]]) + katsuura(z[:, g[1]]) + ackley(z[:, g[2]]) + rastrigin(z[:, g[3]]) + schwefel(z[:, g[4]])
                + schaffer_f7(z[:, :last]))


class F8_CEC2022(CEC2022Function):
    func_num = 8
    parts = (0.3, 0.2, 0.2, 0.1, 0.2)

    def _f(self, x, c):
        z = self._ssr(x, c["o"][0], c["M"][0], 1.0, c["S"])
        g = c["groups"]
        return (katsuura(z[:, g[0]]) + happycat(z[:, g[1]]) + exp_grie_rosen(z[:, g[2]]) + schwefel(z[:, g[3]])
                + ackley(z[:, g[4]]))


class F9_CEC2022(CEC2022Function):
    func_num = 9

    def _f(self, x, c):
        o, M = c["o"], c["M"]
        fs = torch.stack([rosenbrock(self._ssr(x, o[0], M[0], 2.048 / 100.0)), elliptic(self._ssr(x, o[1], M[1])),
                          bent_cigar(self._ssr(x, o[2], M[2])), discus(self._ssr(x, o[3], M[3])), elliptic(x - o[4])], dim=1)
        return self._compose(x, fs, c, [0, 200, 300, 100, 400], [10, 20, 30, 40, 50], [1, 1e-6, 1e-26, 1e-6, 1e-6])


class F10_CEC2022(CEC2022Function):
    func_num = 10

    def _f(self, x, c):
        o, M = c["o"], c["M"]
        fs = torch.stack([schwefel(x - o[0]), rastrigin(self._ssr(x, o[1], M[1])), hgbat(self._ssr(x, o[2], M[2]))], dim=1)
        return self._compose(x, fs, c, [0, 200, 100], [20, 10, 10], [1, 1, 1])


class F11_CEC2022(CEC2022Function):
    func_num = 11

    def _f(self, x, c):
        o, M = c["o"], c["M"]
        fs = torch.stack([exp_schaffer(self._ssr(x, o[0], M[0])), schwefel(self._ssr(x, o[1], M[1])),
                          griewank(self._ssr(x, o[2], M[2], 6.0)), rosenbrock(self._ssr(x, o[3], M[3], 2.048 / 100.0)),
                          rastrigin(self._ssr(x, o[4], M[4]))], dim=1)
        return self._compose(x, fs, c, [0, 200, 300, 400, 200], [20, 20, 30, 30, 20], [5e-4, 1, 10, 1, 10])

    def _floor(self, d: int) -> float:
        return 5.07e-6 if d == 10 else 1.46e-5


class F12_CEC2022(CEC2022Function):
    func_num = 12

    def _f(self, x, c):
        o, M = c["o"], c["M"]
        fs = torch.stack([hgbat(self._ssr(x, o[0], M[0])), rastrigin(self._ssr(x, o[1], M[1])),
                          schwefel(self._ssr(x, o[2], M[2])), bent_cigar(self._ssr(x, o[3], M[3])),
                          elliptic(self._ssr(x, o[4], M[4])), exp_schaffer(self._ssr(x, o[5], M[5]))], dim=1)
        return self._compose(x, fs, c, [0, 300, 500, 100, 400, 200], [10, 20, 30, 40, 50, 60],
                             [10, 10, 2.5, 1e-26, 1e-6, 5e-4])


class CEC2022TestSuit:
    """``CEC2022TestSuit.create(k)`` → an instance of ``Fk_CEC2022``."""

    func_num2class = {1: F1_CEC2022, 2: F2_CEC2022, 3: F3_CEC2022, 4: F4_CEC2022, 5: F5_CEC2022, 6: F6_CEC2022,
                      7: F7_CEC2022, 8: F8_CEC2022, 9: F9_CEC2022, 10: F10_CEC2022, 11: F11_CEC2022, 12: F12_CEC2022}

    @staticmethod
    def create(func_num: int):
        return CEC2022TestSuit.func_num2class[func_num]()

    @staticmethod
    def optimum(func_num: int, dim: int, dtype=torch.float32):
        """Location of the global optimum (first shift vector)."""
        return torch.as_tensor(_archive()[f"shift_{func_num}"][0, :dim]).to(dtype)
