"""MaF1–15 many-objective benchmark suite (Cheng et al., Complex & Intelligent Systems 2017).
Reference: ``src/evox/problems/numerical/maf.py:66-1166``.

All problems evaluate an (N, d) population in one shot.  MaF10–12 are the WFG1/2/9 problems and
are written from the WFG toolkit's transformation functions; the reference's versions of
these index past the last column and mis-slice the reduction groups
(``maf.py:503-541``), so they are re-derived here rather than mirrored.

Decision-space bounds: [0, 1] except MaF8/9 ([-10000, 10000]², polygon problems) and
MaF10–12 (variable i in [0, 2i])."""
from __future__ import annotations

import math

import torch

from ... import _native
from ...core.problem import Problem
from ...core.state import State
from ...operators.sampling import UniformSampling
from ...operators.selection.non_dominate import non_dominated_sort
from ...ops.distance import cdist as _cdist
from ...ops._util import check, hip, on_gpu, stream_of
from .dtlz import _chain, _multimodal_g
from .lsmop import (FRONT_INVERTED, FRONT_LINEAR, _dtlz7_front, _griewank, _rastrigin, _rosenbrock, _sphere,
                    chaotic_group_sizes, fused_grouped, grouped_g)

_native.register_hip_api({"evox_hip_maf": (_native.I, [_native.P, _native.I64, _native.I, _native.I, _native.I, _native.I,
                                                      _native.P, _native.P])})
_native.register_hip_api({"evox_hip_maf9": (_native.I, [_native.P, _native.I64, _native.I, _native.I, _native.P, _native.P,
                                                       _native.P])})
_KIND = {}        # stock class -> variant of the one-pass kernel (``csrc/mo_large.hip``); filled at the end of the module


def _fused(problem, X):
    """Objectives of MaF1-7 / MaF13 from the one-pass kernel when ``problem`` is a stock class and ``X`` an fp32 device
    matrix; None otherwise (subclasses that override pieces use the tensor formulation)."""
    kind = _KIND.get(type(problem))
    if kind is None or not on_gpu(X) or X.dtype != torch.float32 or X.dim() != 2 or problem.m > 16 \
            or X.shape[1] < max(problem.m, 2) or X.shape[0] == 0:
        return None
    if kind in (10, 11, 12) and (X.shape[1] > 64 or (kind == 11 and (X.shape[1] - problem.m + 1) % 2)):
        return None
    if X.stride(1) != 1:
        X = X.contiguous()
    out = torch.empty((X.shape[0], problem.m), dtype=torch.float32, device=X.device)
    check(hip().evox_hip_maf(X.data_ptr(), X.stride(0), X.shape[0], X.shape[1], problem.m, kind, out.data_ptr(),
                             stream_of(X)), "maf")
    return out


# ------------------------------------------------------------------ polygon helpers (MaF8/9)
def inside(x, a, b):
    """x in [min(a,b), max(a,b))."""
    x, a, b = (torch.as_tensor(v, dtype=torch.float32) for v in (x, a, b))
    return (torch.minimum(a, b) <= x) & (x < torch.maximum(a, b))


def ray_intersect_segment(point, seg_init, seg_term):
    """Does the horizontal ray cast from ``point`` towards +x hit the segment?  Broadcasts over
    leading dimensions of its arguments (last dimension = 2)."""
    point, seg_init, seg_term = (torch.as_tensor(v, dtype=torch.float32) for v in (point, seg_init, seg_term))
    px, py = point[..., 0], point[..., 1]
    ax, ay, bx, by = seg_init[..., 0], seg_init[..., 1], seg_term[..., 0], seg_term[..., 1]
    dy = by - ay
    on_flat = (py == ay) & inside(px, ax, bx)
    lhs = ax * dy + (py - ay) * (bx - ax)
    rhs = px * dy
    crosses = ((dy > 0) & (lhs >= rhs)) | ((dy < 0) & (lhs <= rhs))
    return ((dy == 0) & on_flat) | ((dy != 0) & crosses & inside(py, ay, by))


def point_in_polygon(polygon, point):
    """Even–odd rule; vertices count as inside.  ``point`` may be (2,) or (P, 2)."""
    polygon = torch.as_tensor(polygon, dtype=torch.float32)
    point = torch.as_tensor(point, dtype=torch.float32)
    single = point.dim() == 1
    pts = point.reshape(-1, 2)
    nxt = torch.roll(polygon, 1, dims=0)
    hits = ray_intersect_segment(pts[:, None, :], polygon[None, :, :], nxt[None, :, :]).sum(1)
    vertex = (pts[:, None, :] == polygon[None, :, :]).all(-1).any(1)
    out = (hits % 2 == 1) | vertex
    return out[0] if single else out


def regular_polygon(m):
    ang = math.pi / 2 - torch.arange(1, m + 1, dtype=torch.float32) * 2 * math.pi / m
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=1)


# --------------------------------------------------------------------------- WFG toolkit
def s_linear(y, a):
    return (y - a).abs() / (torch.floor(a - y) + a).abs()


def b_flat(y, a, b, c):
    out = (a + torch.clamp_max(torch.floor(y - b), 0) * a * (b - y) / b
           - torch.clamp_max(torch.floor(c - y), 0) * (1 - a) * (y - c) / (1 - c))
    return torch.round(out * 1e4) / 1e4


def b_poly(y, alpha):
    return y.clamp_min(0) ** alpha


def s_decept(y, a, b, c):
    return 1 + ((y - a).abs() - b) * (torch.floor(y - a + b) * (1 - c + (a - b) / b) / (a - b)
                                      + torch.floor(a + b - y) * (1 - c + (1 - a - b) / b) / (1 - a - b) + 1 / b)


def s_multi(y, a, b, c):
    t = (y - c).abs() / 2 / (torch.floor(c - y) + c)
    return (1 + torch.cos((4 * a + 2) * math.pi * (0.5 - t)) + 4 * b * t ** 2) / (b + 2)


def r_sum(y, w):
    return (y * w).sum(1) / w.sum()


def r_nonsep(y, a):
    """Non-separable reduction with degree ``a`` over the columns of ``y``."""
    n = y.shape[1]
    out = y.sum(1)
    for k in range(1, a):
        out = out + (y - torch.roll(y, -k, dims=1)).abs().sum(1)
    return out / (n / a) / math.ceil(a / 2) / (1 + 2 * a - 2 * math.ceil(a / 2))


def shape_convex(x):
    ang = x[:, :-1] * (math.pi / 2)
    return _chain(1 - torch.cos(ang), 1 - torch.sin(ang))


def shape_concave(x):
    ang = x[:, :-1] * (math.pi / 2)
    return _chain(torch.sin(ang), torch.cos(ang))


def shape_mixed(x):
    return 1 - x[:, 0] - torch.cos(10 * math.pi * x[:, 0] + math.pi / 2) / (10 * math.pi)


def shape_disc(x):
    return 1 - x[:, 0] * torch.cos(5 * math.pi * x[:, 0]) ** 2


# --------------------------------------------------------------------------------- suite
class MaF(Problem):
    default_d = None

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__()
        self.m = 3 if m is None else m
        self.d = (self.m + 9 if self.default_d is None else self.default_d) if d is None else d
        self.ref_num = ref_num

    def setup(self, key):
        return State(key=key)

    def _simplex(self, m=None):
        return UniformSampling(self.ref_num * self.m, self.m if m is None else m)()[0]

    def pf(self):
        return 1 - self._simplex()


class MaF1(MaF):
    """Inverted linear front."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = ((X[:, m - 1:] - 0.5) ** 2).sum(1, keepdim=True)
        pos = X[:, : m - 1]
        return (1 + g) - (1 + g) * _chain(pos, 1 - pos), state


class MaF2(MaF):
    """DTLZ2BZ: concave front restricted to the central part, one distance function per objective."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m, d = self.m, X.shape[1]
        step = (d - m + 1) // m
        cols = []
        for i in range(m):
            lo = m - 1 + i * step
            hi = lo + step if i < m - 1 else d
            cols.append((((X[:, lo:hi] / 2 + 0.25) - 0.5) ** 2).sum(1))
        g = torch.stack(cols, dim=1)
        ang = (X[:, : m - 1] / 2 + 0.25) * (math.pi / 2)
        return (1 + g) * _chain(torch.cos(ang), torch.sin(ang)), state

    def pf(self):
        m = self.m
        r = self._simplex()
        c = torch.zeros((r.shape[0], m - 1), dtype=r.dtype)
        for j in range(2, m + 1):
            temp = r[:, j - 1] / r[:, 0] * c[:, m - j + 1: m - 1].prod(1)
            c[:, m - j] = torch.sqrt(1 / (1 + temp ** 2))
        lo, hi = math.cos(3 * math.pi / 8), math.cos(math.pi / 8)
        if m > 5:
            c = c * (hi - lo) + lo
        else:
            c = c[((c >= lo) & (c <= hi)).all(1)]
        return _chain(c, torch.sqrt(1 - c ** 2))


class MaF3(MaF):
    """Convex DTLZ3."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = _multimodal_g(X[:, m - 1:])
        ang = X[:, : m - 1] * (math.pi / 2)
        f = (1 + g) * _chain(torch.cos(ang), torch.sin(ang))
        return torch.cat([f[:, : m - 1] ** 4, f[:, m - 1:] ** 2], dim=1), state

    def pf(self):
        r = self._simplex() ** 2
        t = (torch.sqrt(r[:, :-1]).sum(1) + r[:, -1])[:, None]
        return r / torch.cat([(t ** 2).expand(-1, r.shape[1] - 1), t], dim=1)


class MaF4(MaF):
    """Inverted, badly scaled DTLZ3."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = _multimodal_g(X[:, m - 1:])
        ang = X[:, : m - 1] * (math.pi / 2)
        f = (1 + g) - (1 + g) * _chain(torch.cos(ang), torch.sin(ang))
        return f * 2.0 ** torch.arange(1, m + 1, device=X.device, dtype=X.dtype), state

    def pf(self):
        r = self._simplex()
        r = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
        return (1 - r) * 2.0 ** torch.arange(1, self.m + 1, dtype=r.dtype)


class MaF5(MaF):
    """Badly scaled DTLZ4 (bias α = 100)."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        g = ((X[:, m - 1:] - 0.5) ** 2).sum(1, keepdim=True)
        ang = X[:, : m - 1] ** 100 * (math.pi / 2)
        f = (1 + g) * _chain(torch.cos(ang), torch.sin(ang))
        return f * 2.0 ** torch.arange(m, 0, -1, device=X.device, dtype=X.dtype), state

    def pf(self):
        r = self._simplex()
        r = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
        return r * 2.0 ** torch.arange(self.m, 0, -1, dtype=r.dtype)


class MaF6(MaF):
    """DTLZ5(I, M) with I = 2: a degenerate curve front."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m, i = self.m, 2
        g = ((X[:, m - 1:] - 0.5) ** 2).sum(1, keepdim=True)
        pos = X[:, : m - 1].clone()
        pos[:, i - 1:] = (1 + 2 * g * pos[:, i - 1:]) / (2 + 2 * g)
        ang = pos * (math.pi / 2)
        return (1 + 100 * g) * _chain(torch.cos(ang), torch.sin(ang)), state

    def pf(self):
        m, i = self.m, 2
        r = self._simplex(i)
        r = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
        if r.shape[1] < m:
            r = torch.cat([r[:, :1].expand(-1, m - r.shape[1]), r], dim=1)
        expo = torch.cat([torch.full((1,), float(m - i)), torch.arange(m - i, -1, -1, dtype=r.dtype)]).clamp_min(0)
        return r / math.sqrt(2) ** expo


class MaF7(MaF):
    """DTLZ7: disconnected front."""
    default_d = None

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(d, m, ref_num)
        if d is None:
            self.d = self.m + 19

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
        return _dtlz7_front(self.ref_num * self.m, self.m)


class MaF8(MaF):
    """Multi-point distance minimisation: objective j is the distance to vertex j of a regular
    m-gon; the Pareto set is the polygon itself."""

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(2, m, ref_num)
        self.points = regular_polygon(self.m)

    def evaluate(self, state, X):
        return _cdist(X[:, :2].to(torch.float32), self.points.to(X.device)), state

    def _grid_inside(self):
        side = int(math.ceil(math.sqrt(self.ref_num * self.m)))
        t = torch.linspace(-1, 1, side)
        xx, yy = torch.meshgrid(t, t, indexing="ij")
        pts = torch.stack([xx.reshape(-1), yy.reshape(-1)], dim=1)
        return pts[point_in_polygon(self.points, pts)]

    def pf(self):
        return _cdist(self._grid_inside(), self.points)


class MaF9(MaF8):
    """Multi-line distance minimisation: objective j is the distance to the line through
    vertices j and j+1."""

    def _lines(self, X):
        p = self.points.to(X.device)
        a, b = p, torch.roll(p, -1, dims=0)
        x, y = X[:, :1].to(torch.float32), X[:, 1:2].to(torch.float32)
        num = ((a[:, 0] - x) * (b[:, 1] - y) - (b[:, 0] - x) * (a[:, 1] - y)).abs()
        return num / torch.linalg.vector_norm(a - b, dim=1)

    def evaluate(self, state, X):
        if type(self) is MaF9 and on_gpu(X) and X.dtype == torch.float32 and X.dim() == 2 and X.shape[0] > 0:
            Xc = X if X.stride(1) == 1 else X.contiguous()
            pts = self.points.to(X.device, torch.float32).contiguous()
            out = torch.empty((X.shape[0], self.m), dtype=torch.float32, device=X.device)
            check(hip().evox_hip_maf9(Xc.data_ptr(), Xc.stride(0), Xc.shape[0], self.m, pts.data_ptr(), out.data_ptr(),
                                      stream_of(Xc)), "maf9")
            return out, state
        return self._lines(X), state

    def pf(self):
        return self._lines(self._grid_inside())


class _WFG(MaF):
    def _split(self, X):
        m, d = self.m, X.shape[1]
        k = m - 1
        z = X / torch.arange(2, 2 * d + 1, 2, device=X.device, dtype=X.dtype)
        return z, k, d - k

    def _finish(self, t, shape, last_shape):
        """Degeneracy constants A = 1, distance scale D = 1, objective scales S = 2, 4, …"""
        m = self.m
        x = torch.cat([torch.clamp_min(t[:, m - 1:], 1.0) * (t[:, : m - 1] - 0.5) + 0.5, t[:, m - 1:]], dim=1)
        h = shape(x)
        if last_shape is not None:
            h = torch.cat([h[:, : m - 1], last_shape(x)[:, None]], dim=1)
        s = torch.arange(2, 2 * m + 1, 2, device=t.device, dtype=t.dtype)
        return x[:, m - 1:] + s * h

    def _front_angles(self, extra):
        """Position parameters whose shape-function image hits the sampled directions; the
        first one is found by a 10⁻⁴ grid search on the mixed/disconnected last objective."""
        m = self.m
        R = self._simplex()
        c = torch.ones((R.shape[0], m), dtype=R.dtype)
        for j in range(2, m + 1):
            temp = R[:, j - 1] / R[:, 0] * (1 - c[:, m - j + 1: m - 1]).prod(1)
            c[:, m - j] = (temp ** 2 - temp + torch.sqrt(2 * temp)) / (temp ** 2 + 1)
        x = torch.acos(c.clamp(-1, 1)) * 2 / math.pi
        temp = (1 - torch.sin(math.pi / 2 * x[:, 1])) * R[:, m - 1] / R[:, m - 2]
        a = torch.arange(0, 10001, dtype=R.dtype) / 10000
        best = torch.empty(R.shape[0], dtype=torch.int64)
        for s in range(0, R.shape[0], 512):
            E = (temp[s:s + 512, None] * (1 - torch.cos(math.pi / 2 * a)) - 1 + extra(a)[None, :]).abs()
            best[s:s + 512] = torch.argsort(E, dim=1, stable=True)[:, :10].amin(1)
        x[:, 0] = a[best]
        return x


class MaF10(_WFG):
    """WFG1: polynomial bias and flat region, mixed convex front."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        z, k, l = self._split(X)
        t = torch.cat([z[:, :k], s_linear(z[:, k:], 0.35)], dim=1)
        t = torch.cat([t[:, :k], b_flat(t[:, k:], 0.8, 0.75, 0.85)], dim=1)
        t = b_poly(t, 0.02)
        w = torch.arange(2, 2 * (k + l) + 1, 2, device=X.device, dtype=X.dtype)
        per = k // (m - 1)
        cols = [r_sum(t[:, i * per:(i + 1) * per], w[i * per:(i + 1) * per]) for i in range(m - 1)]
        cols.append(r_sum(t[:, k:], w[k:]))
        return self._finish(torch.stack(cols, dim=1), shape_convex, shape_mixed), state

    def pf(self):
        x = self._front_angles(lambda a: a + torch.cos(10 * math.pi * a + math.pi / 2) / (10 * math.pi))
        f = torch.cat([shape_convex(x)[:, : self.m - 1], shape_mixed(x)[:, None]], dim=1)
        return f * torch.arange(2, 2 * self.m + 1, 2, dtype=f.dtype)


class MaF11(_WFG):
    """WFG2: non-separable reduction, disconnected convex front."""

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(d, m, ref_num)
        self.d = int(math.ceil((self.d - self.m + 1) / 2) * 2 + self.m - 1)

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        z, k, l = self._split(X)
        t = torch.cat([z[:, :k], s_linear(z[:, k:], 0.35)], dim=1)
        a, b = t[:, k::2], t[:, k + 1::2]
        pairs = min(a.shape[1], b.shape[1])
        a, b = a[:, :pairs], b[:, :pairs]
        t = torch.cat([t[:, :k], (a + b + 2 * (a - b).abs()) / 3], dim=1)
        per = k // (m - 1)
        cols = [t[:, i * per:(i + 1) * per].mean(1) for i in range(m - 1)]
        cols.append(t[:, k:].mean(1))
        return self._finish(torch.stack(cols, dim=1), shape_convex, shape_disc), state

    def pf(self):
        x = self._front_angles(lambda a: a * torch.cos(5 * math.pi * a) ** 2)
        f = torch.cat([shape_convex(x)[:, : self.m - 1], shape_disc(x)[:, None]], dim=1)
        f = f[non_dominated_sort(f) == 0]
        return f * torch.arange(2, 2 * self.m + 1, 2, dtype=f.dtype)


class MaF12(_WFG):
    """WFG9: parameter-dependent bias, deceptive and multi-modal shifts, non-separable
    reductions, concave front."""

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        m = self.m
        z, k, l = self._split(X)
        n_var = k + l
        tail_sum = torch.flip(torch.cumsum(torch.flip(z, [1]), 1), [1]) - z
        cnt = torch.arange(n_var - 1, -1, -1, device=X.device, dtype=X.dtype)
        Y = tail_sum[:, :-1] / cnt[:-1]
        expo = 0.02 + (50 - 0.02) * (0.98 / 49.98 - (1 - 2 * Y) * (torch.floor(0.5 - Y) + 0.98 / 49.98).abs())
        t = torch.cat([z[:, :-1].clamp_min(0) ** expo, z[:, -1:]], dim=1)
        t = torch.cat([s_decept(t[:, :k], 0.35, 0.001, 0.05), s_multi(t[:, k:], 30, 95, 0.35)], dim=1)
        per = k // (m - 1)
        cols = [r_nonsep(t[:, i * per:(i + 1) * per], per) for i in range(m - 1)]
        tail = t[:, k:]
        pair_sum = (tail[:, :, None] - tail[:, None, :]).abs().sum((1, 2)) / 2
        cols.append((tail.sum(1) + 2 * pair_sum) / math.ceil(l / 2) / (1 + 2 * l - 2 * math.ceil(l / 2)))
        return self._finish(torch.stack(cols, dim=1), shape_concave, None), state

    def pf(self):
        r = self._simplex()
        r = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
        return r * torch.arange(2, 2 * self.m + 1, 2, dtype=r.dtype)


class MaF13(MaF):
    """PF7: three conflicting objectives plus m−3 objectives dependent on them; non-linear
    variable linkage."""
    default_d = 5

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(d, m, ref_num)
        self.m = max(self.m, 3)

    def evaluate(self, state, X):
        f = _fused(self, X)
        if f is not None:
            return f, state
        n, d = X.shape
        j = torch.arange(1, d + 1, device=X.device, dtype=X.dtype)
        Y = X - 2 * X[:, 1:2] * torch.sin(2 * math.pi * X[:, :1] + j * math.pi / d)
        a, b = X[:, 0] * (math.pi / 2), X[:, 1] * (math.pi / 2)

        def mean_sq(cols):
            return (cols ** 2).mean(1) if cols.shape[1] > 0 else torch.zeros(n, device=X.device, dtype=X.dtype)

        f1 = torch.sin(a) + 2 * mean_sq(Y[:, 3:d:3])
        f2 = torch.cos(a) * torch.sin(b) + 2 * mean_sq(Y[:, 4:d:3])
        f3 = torch.cos(a) * torch.cos(b) + 2 * mean_sq(Y[:, 2:d:3])
        rest = f1 ** 2 + f2 ** 10 + f3 ** 10 + 2 * mean_sq(Y[:, 3:d])
        return torch.cat([torch.stack([f1, f2, f3], dim=1), rest[:, None].expand(-1, self.m - 3)], dim=1), state

    def pf(self):
        r = self._simplex(3)
        r = r / torch.linalg.vector_norm(r, dim=1, keepdim=True)
        rest = (r[:, 0] ** 2 + r[:, 1] ** 10 + r[:, 2] ** 10)[:, None].expand(-1, self.m - 3)
        return torch.cat([r, rest], dim=1)


class _LargeScaleMaF(MaF):
    nk = 2

    def __init__(self, d=None, m=None, ref_num=1000):
        super().__init__(d, m, ref_num)
        if d is None:
            self.d = 20 * self.m
        self.sublen = tuple(chaotic_group_sizes(self.m, self.d - self.m + 1, self.nk))


class MaF14(_LargeScaleMaF):
    """LSMOP3: linear front, linear linkage, Rastrigin / Rosenbrock group landscapes."""

    def evaluate(self, state, X):
        if type(self) is MaF14:
            f = fused_grouped(X, self.m, self.nk, self.sublen, (_rastrigin, _rosenbrock), False, FRONT_LINEAR)
            if f is not None:
                return f, state
        m, d = self.m, X.shape[1]
        w = 1 + torch.arange(m, d + 1, device=X.device, dtype=X.dtype) / d
        X = torch.cat([X[:, : m - 1], w * X[:, m - 1:] - X[:, :1] * 10], dim=1)
        g = grouped_g(X, m, self.sublen, self.nk, (_rastrigin, _rosenbrock))
        pos = X[:, : m - 1]
        return (1 + g) * _chain(pos, 1 - pos), state

    def pf(self):
        return self._simplex()


class MaF15(_LargeScaleMaF):
    """Inverted LSMOP8: convex front, non-linear linkage, Griewank / sphere landscapes."""

    def evaluate(self, state, X):
        if type(self) is MaF15:
            f = fused_grouped(X, self.m, self.nk, self.sublen, (_griewank, _sphere), True, FRONT_INVERTED)
            if f is not None:
                return f, state
        m, d = self.m, X.shape[1]
        w = 1 + torch.cos(torch.arange(m, d + 1, device=X.device, dtype=X.dtype) / d * (math.pi / 2))
        X = torch.cat([X[:, : m - 1], w * X[:, m - 1:] - X[:, :1] * 10], dim=1)
        g = grouped_g(X, m, self.sublen, self.nk, (_griewank, _sphere))
        gg = 1 + g + torch.cat([g[:, 1:], torch.zeros_like(g[:, :1])], dim=1)
        ang = X[:, : m - 1] * (math.pi / 2)
        return gg * (1 - _chain(torch.cos(ang), torch.sin(ang))), state

    def pf(self):
        r = self._simplex()
        return 1 - r / torch.linalg.vector_norm(r, dim=1, keepdim=True)


_KIND.update({MaF1: 1, MaF2: 2, MaF3: 3, MaF4: 4, MaF5: 5, MaF6: 6, MaF7: 7, MaF10: 10, MaF11: 11, MaF12: 12, MaF13: 13})
