"""CMA-ES hot path on the bf16 matrix cores (kernels: ``csrc/bf16_gemm.h``, ``csrc/cma_bf16.hip``, ``csrc/blas.hip``).

Every GEMM-shaped step of a generation is the same hand-written ``C = A·Bᵀ`` kernel with a different epilogue:

``sample``        ``X = m + σ·Z·Sᵀ`` with ``Z`` bf16 Gaussian draws and ``S`` the bf16 copy of the sampling factor
                  (reference formula: ``src/evox/algorithms/so/es_variants/cma_es.py:128-135``).
``elite_factor_t`` gathers the selected rows, centres, scales by ``√w/σ`` and stores them TRANSPOSED
                  (``Yt[c, i]``), so that the rank-μ term is an ``A·Aᵀ`` product with contiguous K.
``covariance``    ``C' = αC + c₁·pc pcᵀ + cμ·Yt·Ytᵀ`` (``cma_es.py:178-184``): upper-triangular tiles only,
                  K split over workgroups, summed in a fixed order and mirrored.
``ns_sqrt``       coupled Newton–Schulz iteration for ``C^{±1/2}`` on symmetric products of bf16 split planes.

All bf16 operands live in row-padded buffers (pitch a multiple of 64 elements, pad columns zero) owned by a
``Workspace`` that is allocated once per problem size, so a generation allocates nothing but its outputs.
"""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I, I64, F = _native.P, _native.I, _native.I64, _native.F
_native.register_hip_api({
    "evox_hip_normal_bf16_pitched": (I, [P, P, I64, I64, I64, I64, P]),
    "evox_hip_bf16_sample": (I, [P, I64, P, I64, P, P, P, I64, I, I, I, P]),
    "evox_hip_cma_gather_t": (I, [P, I64, P, P, P, P, P, I64, P, I, P, I, I, I, P]),
    "evox_hip_bf16_cov_splits": (I, [I, I]),
    "evox_hip_bf16_cov": (I, [P, I64, I, I, P, P, P, F, F, P, P, I64, P]),
    "evox_hip_bf16_gram_packed": (I, [P, I64, I, I, P, P, I64, P]),
    "evox_hip_cov_from_packed": (I, [P, I, P, P, P, F, F, P, P, P]),
    "evox_hip_ns_scal_floats": (I, []),
    "evox_hip_ns_sqrt_planes": (I, [P, I, I64, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P]),
    "evox_hip_matvec": (I, [P, I, P, P, P]),
})


def pitch(k: int) -> int:
    """Row pitch (elements) of a bf16 operand with ``k`` columns: whole 64-element steps, an odd number of them
    (consecutive rows then start in different memory channels)."""
    return (((int(k) + 63) // 64) | 1) * 64


def available(t: torch.Tensor) -> bool:
    return t.is_cuda


def padded_bf16(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """``x`` (rows, k) as a zero-padded bf16 buffer of pitch ``pitch(k)``; returns the (rows, pitch) buffer."""
    rows, k = x.shape
    if out is None:
        out = torch.zeros((rows, pitch(k)), dtype=torch.bfloat16, device=x.device)
    out[:, :k].copy_(x)
    return out


class Workspace:
    """Device buffers of one CMA-ES instance (dimension ``n``, ``lam`` local rows per generation, ``mu`` rows
    entering the covariance update).  ``planes`` of the Newton–Schulz driver are created on first use."""

    def __init__(self, n: int, lam: int, mu: int, device, ns_planes: int = 2):
        self.n, self.lam, self.mu, self.device = int(n), int(lam), int(mu), device
        self.ldn, self.ldmu = pitch(n), pitch(mu)
        # Gaussian draws of a generation: two buffers, so that the draws of the next generation can be produced on
        # a side stream while the current one is still being consumed (``prefetch_normal``)
        self._Z = [torch.zeros((lam, self.ldn), dtype=torch.bfloat16, device=device) for _ in range(2)]
        self._z_next = 0
        self._prefetch = None            # (key tensor, rows, row_offset, buffer, event)
        self._side = None
        self.Yt = torch.zeros((n, self.ldmu), dtype=torch.bfloat16, device=device)
        self.rb = max(1, min(40, (mu + 63) // 64))          # row-tile groups of the gather that run as separate workgroups (their partial sums are added up later)
        self.colsum = torch.empty((self.rb, n), dtype=torch.float32, device=device)
        self.cov_splits = int(hip().evox_hip_bf16_cov_splits(n, mu)) if (n % 4 == 0) else 0
        self.cov_ws = torch.empty((max(self.cov_splits, 1), n, n), dtype=torch.float32, device=device) if self.cov_splits else None
        self.zero = torch.zeros((), dtype=torch.float32, device=device)
        self.ns_planes = int(ns_planes)
        self._ns = None
        # bf16 images of the sampling factor: a ring of two, handed out alternately (see ``factor_buffer``)
        self._factor = [torch.zeros((n, self.ldn), dtype=torch.bfloat16, device=device) for _ in range(2)]
        self._factor_next = 0

    @property
    def Z(self) -> torch.Tensor:
        return self._Z[self._z_next]

    def draws(self, key: torch.Tensor, rows: int, n: int, row_offset: int) -> torch.Tensor:
        """The (rows, pitch) bf16 draws ``normal(key, (·, n))[row_offset:row_offset + rows]``: taken from the
        prefetch when it was made for this very key tensor, drawn on the current stream otherwise."""
        pf, self._prefetch = self._prefetch, None
        if pf is not None and pf[0] is key and pf[1] == rows and pf[2] == row_offset:
            torch.cuda.current_stream(self.device).wait_event(pf[4])
            self._z_next = 0 if pf[3] is self._Z[0] else 1
            return pf[3][:rows]
        z = self._Z[self._z_next][:rows]
        normal_pitched(key, z, n, row_offset)
        return z

    def prefetch_normal(self, key: torch.Tensor, rows: int, n: int, row_offset: int):
        """Start the draws of the NEXT generation (its sampling key is already known) on a side stream into the
        buffer that is not in use; ``draws`` hands them out when asked for the same key tensor."""
        if torch.cuda.is_current_stream_capturing():
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        buf = self._Z[self._z_next ^ 1]
        self._side.wait_stream(main)             # the key exists, and the last reader of `buf` has been issued
        with torch.cuda.stream(self._side):
            normal_pitched(key, buf[:rows], n, row_offset)
            ev = torch.cuda.Event()
            ev.record(self._side)
        key.record_stream(self._side)
        self._prefetch = (key, rows, row_offset, buf, ev)

    def factor_buffer(self) -> torch.Tensor:
        """A zero-padded (n, pitch(n)) bf16 buffer for the next sampling factor.  Two buffers alternate, so the
        factor of the previous generation stays intact while the next one is written; whoever caches a buffer must
        key it by the fp32 tensor it was made from (``CMAES._factor_bf16`` does)."""
        buf = self._factor[self._factor_next]
        self._factor_next ^= 1
        return buf

    def ns_planes_buffer(self, planes: int) -> torch.Tensor:
        """Zero-filled plane storage of the factorisation for ``planes`` (2 or 3) planes per matrix.  Each count has a
        buffer of its own: the pad columns of a plane are never written and must stay zero, and the two layouts put
        them in different places.  The three-plane buffer (ill-conditioned covariances) is allocated on first use."""
        if planes == self.ns_planes:
            return self.ns()["planes"]
        extra = self.__dict__.setdefault("_ns_extra", {})
        if planes not in extra:
            extra[planes] = torch.zeros((10 * planes, self.n, self.ldn), dtype=torch.bfloat16, device=self.device)
        return extra[planes]

    def ns(self):
        if self._ns is None:
            n, dev = self.n, self.device
            g = torch.Generator().manual_seed(0x5EED)
            v = torch.randn(n, generator=g)
            vec = torch.zeros(2 * n, dtype=torch.float32, device=dev)
            vec[:n] = (v / v.norm()).to(dev)
            self._ns = {
                "planes": torch.zeros((10 * self.ns_planes, n, self.ldn), dtype=torch.bfloat16, device=dev),
                "f32": torch.empty((3, n, n), dtype=torch.float32, device=dev),
                "vec": vec,
                "scal": torch.zeros(int(hip().evox_hip_ns_scal_floats()), dtype=torch.float32, device=dev),
            }
        return self._ns


def normal_pitched(key: torch.Tensor, out: torch.Tensor, n: int, row_offset: int = 0) -> torch.Tensor:
    """Fill ``out[:, :n]`` (bf16, rows × pitch) with the draws ``normal(key, (·, n))[row_offset : row_offset + rows]``
    rounded to bf16: bit-identical to the unpadded draw followed by a cast."""
    rows, ld = out.shape
    check(hip().evox_hip_normal_bf16_pitched(key.contiguous().data_ptr(), out.data_ptr(), rows, int(n), ld, int(row_offset),
                                             stream_of(out)), "normal_bf16_pitched")
    return out


def sample(key, mean, sigma, S_bf16, pop_size: int, ws: Workspace, row_offset: int = 0, out=None):
    """``pop_size`` rows of the generation's sample starting at logical row ``row_offset``.
    ``S_bf16``: (n, pitch(n)) bf16 sampling factor with zero pad columns (``S·Sᵀ = C``)."""
    n = mean.shape[0]
    if pop_size > ws.lam or n != ws.n or S_bf16.shape != (n, ws.ldn) or S_bf16.dtype != torch.bfloat16:
        raise ValueError("sample: operands do not match the workspace")
    z = ws.draws(key, pop_size, n, row_offset)
    if out is None:
        out = torch.empty((pop_size, n), dtype=torch.float32, device=mean.device)
    sig = sigma.reshape(1) if isinstance(sigma, torch.Tensor) else torch.tensor([float(sigma)], device=mean.device)
    mean = mean.contiguous()
    if n % 4 or mean.data_ptr() % 16:
        raise ValueError("sample: the dimension must be a multiple of 4 and the mean 16-byte aligned")
    check(hip().evox_hip_bf16_sample(z.data_ptr(), ws.ldn, S_bf16.data_ptr(), ws.ldn, mean.data_ptr(), sig.data_ptr(),
                                     out.data_ptr(), n, pop_size, n, n, stream_of(mean)), "bf16_sample")
    return out


def elite_factor_t(pop, order, mean, sigma, rowscale, ws: Workspace, out=None):
    """Writes ``ws.Yt[c, i] = rowscale[i]·(pop[order[i], c] − mean[c])/σ`` for ``i < len(rowscale)`` and returns the
    weighted mean shift ``Σ rowscale[i]²·(pop[order[i]] − mean)`` (fp32, fixed summation order; into ``out``, a
    contiguous fp32 vector of n elements, when given)."""
    mu, n = rowscale.shape[0], mean.shape[0]
    if mu != ws.mu or n != ws.n:
        raise ValueError("elite_factor_t: operands do not match the workspace")
    if pop.stride(1) != 1:
        pop = pop.contiguous()
    if order.dtype != torch.int64:
        order = order.to(torch.int64)
    if order.shape[0] < mu:
        raise ValueError("order is shorter than the number of selected rows")
    sig = sigma.reshape(1) if isinstance(sigma, torch.Tensor) else torch.tensor([float(sigma)], device=mean.device)
    if out is not None and not (out.dtype == torch.float32 and out.is_contiguous() and out.shape == (n,)):
        raise ValueError("elite_factor_t: out must be a contiguous fp32 vector of n elements")
    step = torch.empty(n, dtype=torch.float32, device=mean.device) if out is None else out
    check(hip().evox_hip_cma_gather_t(pop.data_ptr(), pop.stride(0), order.contiguous().data_ptr(), mean.contiguous().data_ptr(),
                                      rowscale.contiguous().data_ptr(), sig.data_ptr(), ws.Yt.data_ptr(), ws.ldmu,
                                      ws.colsum.data_ptr(), ws.rb, step.data_ptr(), mu, n, pop.shape[0], stream_of(pop)),
          "cma_gather_t")
    return step


def covariance(C, pc, alpha, c1: float, cmu: float, ws: Workspace, out=None):
    """``αC + c₁·pc pcᵀ + cμ·Yt·Ytᵀ`` from the factor left in ``ws.Yt`` (``alpha`` a 0-d device tensor)."""
    n = ws.n
    if not ws.cov_splits:
        raise ValueError("covariance: the dimension must be a multiple of 4")
    if out is None:
        out = torch.empty((n, n), dtype=torch.float32, device=C.device)
    check(hip().evox_hip_bf16_cov(ws.Yt.data_ptr(), ws.ldmu, n, ws.mu, C.contiguous().data_ptr(), pc.contiguous().data_ptr(),
                                  alpha.reshape(1).data_ptr(), float(c1), float(cmu), out.data_ptr(), ws.cov_ws.data_ptr(),
                                  ws.cov_ws.numel(), stream_of(C)), "bf16_cov")
    return out


def packed_upper_floats(n: int) -> int:
    """Floats of the tile-packed upper triangle of an n × n matrix: its 64 × 64 tiles (i ≤ j), each contiguous."""
    t = (n + 63) // 64
    return t * (t + 1) // 2 * 4096


def gram_packed(ws: Workspace, out: torch.Tensor):
    """Upper triangle of ``Yt·Ytᵀ`` into ``out`` (flat fp32, ``packed_upper_floats(n)`` elements, 16-byte aligned), tile
    by tile: what a multi-process job sums across ranks instead of the full square (0.56 n² floats at n = 1000)."""
    if out.dim() != 1 or out.numel() != packed_upper_floats(ws.n) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("gram_packed: out must be a flat fp32 vector of packed_upper_floats(n) elements")
    check(hip().evox_hip_bf16_gram_packed(ws.Yt.data_ptr(), ws.ldmu, ws.n, ws.mu, out.data_ptr(), ws.cov_ws.data_ptr(),
                                          ws.cov_ws.numel(), stream_of(out)), "bf16_gram_packed")
    return out


def unpack_upper(packed: torch.Tensor, n: int) -> torch.Tensor:
    """The symmetric (n, n) matrix of a tile-packed upper triangle (tensor ops; for tests and the op-by-op path)."""
    t = (n + 63) // 64
    full = packed.new_zeros((t * 64, t * 64))
    k = 0
    for i in range(t):
        for j in range(i, t):
            tile = packed[k * 4096:(k + 1) * 4096].view(64, 64)
            full[i * 64:(i + 1) * 64, j * 64:(j + 1) * 64] = tile
            if i != j:
                full[j * 64:(j + 1) * 64, i * 64:(i + 1) * 64] = tile.T
            k += 1
    full = full[:n, :n]
    return torch.triu(full) + torch.triu(full, 1).T


def gram(ws: Workspace, out: torch.Tensor):
    """``Yt·Ytᵀ`` alone (both triangles) into ``out`` (n × n fp32, 16-byte aligned)."""
    n = ws.n
    check(hip().evox_hip_bf16_cov(ws.Yt.data_ptr(), ws.ldmu, n, ws.mu, out.data_ptr(), ws.colsum.data_ptr(),
                                  ws.zero.reshape(1).data_ptr(), 0.0, 1.0, out.data_ptr(), ws.cov_ws.data_ptr(),
                                  ws.cov_ws.numel(), stream_of(out)), "bf16_gram")
    return out


def ns_sqrt(C, ws: Workspace, iters: int, scal: torch.Tensor | None = None, vec: torch.Tensor | None = None, *,
            power_steps: int = 1, safe: bool = False, prev=None, full: bool = False, planes: int | None = None):
    """``(sqrtC, invsqrtC, sqrtC_bf16)`` after ``iters`` (≥ 1) coupled Newton–Schulz steps whose products run on bf16
    split planes: symmetric-half products mirrored on write (well-conditioned ``C`` only), or with ``full`` plain
    products in the iterations after the first step (any conditioning: see ``NewtonSchulzSqrt.step``).  ``planes``
    (default: the workspace's count): 2 carries 16 significant bits per operand, 3 carries 24.
    ``sqrtC_bf16`` is the (n, pitch(n)) zero-padded bf16 copy the sampling product reads.  ``scal`` (the driver's
    scalar block) and ``vec`` (2n floats, the first n a unit vector carried between calls) default to the
    workspace's own; ``scal[13:16]`` afterwards holds the scale, the residual bound and the flag telling that
    ``prev`` was passed through."""
    n = ws.n
    if C.dtype != torch.float32 or tuple(C.shape) != (n, n):
        raise ValueError("ns_sqrt expects an fp32 matrix matching the workspace")
    st = ws.ns()
    scal = st["scal"] if scal is None else scal
    vec = st["vec"] if vec is None else vec
    C = C.contiguous()
    out_s, out_i = torch.empty_like(C), torch.empty_like(C)
    out_b = ws.factor_buffer()
    ps, pi = (prev[0].contiguous(), prev[1].contiguous()) if prev is not None else (None, None)
    planes = ws.ns_planes if planes is None else int(planes)
    check(hip().evox_hip_ns_sqrt_planes(C.data_ptr(), n, ws.ldn, planes, max(1, int(iters)), int(power_steps),
                                      int(bool(safe)), int(bool(full)), ws.ns_planes_buffer(planes).data_ptr(), st["f32"].data_ptr(), vec.data_ptr(),
                                      scal.data_ptr(), ps.data_ptr() if ps is not None else None,
                                      pi.data_ptr() if pi is not None else None, out_s.data_ptr(), out_i.data_ptr(),
                                      out_b.data_ptr(), stream_of(C)), "ns_sqrt_bf16")
    return out_s, out_i, out_b


def matvec(A: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``A @ x`` for a contiguous fp32 (n, n) matrix: one wavefront per row, a fixed summation order (the whitening
    product of the path update; the native generation driver runs the same kernel)."""
    n = A.shape[0]
    y = torch.empty(n, dtype=torch.float32, device=A.device)
    check(hip().evox_hip_matvec(A.contiguous().data_ptr(), n, x.contiguous().data_ptr(), y.data_ptr(), stream_of(A)), "matvec")
    return y
