"""The three heavy steps of covariance-matrix adaptation.

``sample``            X = m + σ · (Z ∘ d) · Bᵀ with Z ~ N(0, I) drawn from the Philox stream
                      of ``key`` inside the GEMM (Z is never written to memory).
``covariance_update`` C' = α·C + c₁·pc pcᵀ + cμ · Yᵀ diag(w) Y with Y gathered from the
                      population through the fitness order, centred and scaled on the fly.
``eigh``              symmetric eigendecomposition C = B diag(d²) Bᵀ (+ C^{-1/2}).

Each has a PyTorch reference (host path / numerics oracle) and a gfx950 implementation.
Reference formulas: ``src/evox/algorithms/so/es_variants/cma_es.py:128-196``.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of


# --------------------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------------------
def sample_reference(key, mean, sigma, B, D, pop_size: int, row_offset: int = 0):
    n = mean.shape[0]
    z = rnd.normal(key, (pop_size, n), offset=row_offset * n)
    return mean + sigma * ((z if D is None else z * D) @ B.T)


def sample(key, mean, sigma, B, D, pop_size: int, precision: str = "fp32", row_offset: int = 0):
    """``pop_size`` rows starting at logical row ``row_offset`` of the generation's sample.
    ``D=None`` means ``B`` already is the full factor (``B Bᵀ = C``), saving the column scaling."""
    if on_gpu(mean):
        from . import gemm

        return gemm.cma_sample(key, mean, sigma, B, D, pop_size, precision, row_offset)
    return sample_reference(key, mean, sigma, B, D, pop_size, row_offset)


# --------------------------------------------------------------------------------------
# rank-mu covariance update
# --------------------------------------------------------------------------------------
def covariance_update_reference(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu):
    mu = weights.shape[0]
    y = (pop[order[:mu]] - mean) / sigma
    return alpha * C + c1 * torch.outer(pc, pc) + cmu * ((y.T * weights) @ y)


def covariance_update(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu, precision: str = "fp32"):
    """``alpha`` may be a 0-d tensor (it depends on ``hsig``)."""
    if on_gpu(C):
        from . import gemm

        return gemm.cma_covariance_update(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu, precision)
    return covariance_update_reference(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu)


def use_factor_form() -> bool:
    """GPU covariance update through the explicit factor S (gather kernel + library GEMM), the
    fastest measured form; ``EVOX_AMD_CMA_COV=fused`` selects the hand-written gathered-operand
    MFMA kernel instead."""
    import os

    return os.environ.get("EVOX_AMD_CMA_COV", "factor") == "factor" and os.environ.get("EVOX_AMD_GEMM", "hip") != "torch"


def elite_factor(pop, order, mean, sigma, sqrt_w, precision: str = "fp32"):
    from . import gemm

    return gemm.cma_elite_factor(pop, order, mean, sigma, sqrt_w, precision)


def covariance_from_factor(C, S, pc, alpha, c1, cmu):
    from . import gemm

    return gemm.cma_covariance_from_factor(C, S, pc, alpha, c1, cmu)


# --------------------------------------------------------------------------------------
# eigendecomposition
# --------------------------------------------------------------------------------------
def _host_max() -> int:
    """Largest matrix order whose dense factorisations take the host round trip (``EVOX_AMD_HOST_LINALG_MAX``).
    Measured on an MI355X box: the device solvers are latency-bound for small matrices — a symmetric
    eigendecomposition takes 1.35 ms at n = 100 on the GPU against 0.30 ms for copy to the host, LAPACK and copy
    back (n = 300: 5.5 vs 2.8 ms; even at n = 500), a thin SVD 2.6 vs 0.43 ms."""
    return int(os.environ.get("EVOX_AMD_HOST_LINALG_MAX", "384"))


def eigh_reference(C) -> Tuple[torch.Tensor, torch.Tensor]:
    C = torch.triu(C) + torch.triu(C, 1).T
    if C.is_cuda and C.shape[-1] <= _host_max():
        evals, B = torch.linalg.eigh(C.cpu())
        return B.to(C.device, non_blocking=True), evals.to(C.device, non_blocking=True)
    evals, B = torch.linalg.eigh(C)
    return B, evals


def small_svd_vt(X: torch.Tensor) -> torch.Tensor:
    """``Vᵀ`` of the thin SVD of ``X``; small problems on a GPU go through the host (see ``_host_max``)."""
    if X.is_cuda and max(X.shape) <= _host_max():
        return torch.linalg.svd(X.cpu(), full_matrices=False)[2].to(X.device, non_blocking=True)
    return torch.linalg.svd(X, full_matrices=False)[2]


def decompose(C, B_prev: Optional[torch.Tensor] = None, method: Optional[str] = None):
    """``(B, D, invsqrtC)`` with ``C = B diag(D²) Bᵀ`` and ``invsqrtC = B diag(1/D) Bᵀ``.

    method: "jacobi"  – warm-started block-Jacobi solver written for gfx950 (GPU default),
            "solver"  – the vendor dense solver behind ``torch.linalg.eigh``.
    ``B_prev`` (the previous generation's eigenvectors) warm-starts the Jacobi solver."""
    method = method or os.environ.get("EVOX_AMD_EIGH", "auto")
    if on_gpu(C) and method in ("auto", "jacobi"):
        from . import eigh as _eigh

        if _eigh.available():
            B, evals = _eigh.jacobi_eigh(C, B_prev)
        elif method == "jacobi":
            raise RuntimeError("the Jacobi eigensolver is not built")
        else:
            B, evals = eigh_reference(C)
    else:
        B, evals = eigh_reference(C)
    D = torch.sqrt(evals.clamp_min(1e-30))
    invsqrtC = (B / D) @ B.T
    return B, D, invsqrtC


# --------------------------------------------------------------------------------------
# matrix square roots without an eigendecomposition
# --------------------------------------------------------------------------------------
class NewtonSchulzSqrt:
    """``C^{1/2}`` and ``C^{-1/2}`` of a symmetric positive definite matrix by the coupled
    Newton–Schulz iteration (GEMMs only, quadratically convergent)::

        Y₀ = C / s,  Z₀ = I,   T = (3I − Z Y) / 2,   Y ← Y T,   Z ← T Z
        Y → (C/s)^{1/2},  Z → (C/s)^{-1/2}          (s = ‖C‖∞ ≥ λmax)

    The number of iterations adapts to the conditioning of ``C`` without stalling the GPU:
    every call leaves its final residual ``‖I − Z Y‖_F / √n`` in pinned host memory through an
    asynchronous copy, and the *next* call reads whatever residual has already arrived and
    raises or lowers its iteration count.  Since the covariance drifts slowly between
    generations the controller tracks the need with one generation of lag."""

    def __init__(self, tol: float = 2e-6, min_iters: int = 2, max_iters: int = 40, start_iters: int = 6,
                 bf16_copy: bool = False):
        self.tol, self.min_iters, self.max_iters = tol, min_iters, max_iters
        self.bf16_copy = bf16_copy        # also emit the square root rounded to bf16 (native driver only)
        self.iters = start_iters
        self._pending = None          # (event, host residual record, iters used)
        self.last_residual = None
        self._graph_residual = None
        self._host = None
        self._event = None
        self._eye = None
        self._ws = None               # native driver scratch (GPU)
        self.safe_calls = 0           # > 0: scale by the norm bound instead of the λmax estimate
        self.fallbacks = 0            # times the guarded outputs passed the previous factors on

    def _poll(self):
        if self._pending is None:
            return
        ev, host, used = self._pending
        if ev is not None and not ev.query():
            return
        self._pending = None
        r = float(host[0])
        failed = host.numel() > 1 and float(host[1]) != 0.0
        self.last_residual = r
        if failed or not (r == r) or r >= 0.25:
            # the estimate-scaled iteration did not contract: go back to the provably
            # convergent scaling for a while (it needs more iterations)
            self.fallbacks += int(failed)
            self.safe_calls = 64
        # convergence is quadratic: one more iteration squares the residual, one fewer takes its
        # root — raise by as many as the tolerance needs, lower only when the root still passes
        if not (r == r) or r > self.tol:
            more = 1 if r * r < self.tol else (2 if r < 0.19 else 4)
            self.iters = min(self.max_iters, max(self.iters, used) + (more if r == r else 4))
        elif r ** 0.5 < 0.25 * self.tol and used <= self.iters:
            self.iters = max(self.min_iters, used - 1)

    def needs_more_iterations(self) -> bool:
        """For recorded (graph-replayed) use: read the residual the last replay left on the
        device (one synchronisation) and raise the iteration count if it misses the tolerance."""
        if self._graph_residual is None:
            return False
        r = float(self._graph_residual)
        self.last_residual = r
        if not (r == r) or r > self.tol:
            self.iters = min(self.max_iters, self.iters + (4 if (r != r or r > 1e-2) else 2))
            return True
        return False

    def _native_ok(self, C) -> bool:
        if not (C.is_cuda and C.dtype == torch.float32 and os.environ.get("EVOX_AMD_NS", "native") == "native"):
            return False
        from . import blas as _blas
        return _blas.available()

    def _call_native(self, C, prev, capturing):
        from . import blas as _blas
        n = C.shape[0]
        if self._ws is None or self._ws.n != n or self._ws.work.device != C.device:
            self._ws = _blas.NSWorkspace(n, C.device)
        safe = self.safe_calls > 0
        if safe and not capturing:
            self.safe_calls -= 1
        out = _blas.ns_sqrt(C, self._ws, max(1, self.iters), power_steps=1, safe=safe, prev=prev, bf16_copy=self.bf16_copy)
        if capturing:
            self._graph_residual = self._ws.scal[14]
        elif self._pending is None:
            if self._host is None or self._host.numel() != 2:
                self._host = torch.empty(2, dtype=torch.float32, pin_memory=True)
                self._event = torch.cuda.Event()
            self._host.copy_(self._ws.scal[14:16], non_blocking=True)
            self._event.record()
            self._pending = (self._event, self._host, self.iters)
        return out

    def __call__(self, C: torch.Tensor, prev=None):
        """``prev``: the previous ``(sqrtC, invsqrtC)``; the GPU driver passes them on when an
        iteration fails to contract instead of returning garbage."""
        n = C.shape[0]
        gpu = C.is_cuda
        capturing = gpu and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self._poll()
        if self._native_ok(C):
            return self._call_native(C, prev, capturing)
        eye = self._eye
        if eye is None or eye.shape[0] != n or eye.device != C.device or eye.dtype != C.dtype:
            eye = self._eye = torch.eye(n, device=C.device, dtype=C.dtype)
        # scale towards the centre of the spectrum when that is provably safe (λmax < 2s needs
        # ‖C‖∞ < 2s), else by the norm bound: a centred spectrum saves iterations
        s = torch.maximum(torch.diagonal(C).mean(), C.abs().sum(1).max() / 1.9)
        Y = C / s
        # first iteration with Z₀ = I written out: T = (3I − Y)/2, Z₁ = T (saves two GEMMs)
        T = torch.add(eye, Y, alpha=-0.5).add_(eye, alpha=0.5) if self.iters > 0 else eye
        Z = T
        if self.iters > 0:
            Y = Y @ T
        # the multiplication ORDER below is the numerically stable coupled form; products are
        # plain dense GEMMs and go to the vendor library (rocBLAS/hipBLASLt through torch)
        for _ in range(self.iters - 1):
            T = torch.addmm(eye, Z, Y, beta=1.5, alpha=-0.5)
            Y = Y @ T
            Z = T @ Z
        # I − Z Y evaluated before the last update equals 2 (I − T): one iteration later the
        # residual is its square (quadratic convergence), so this bounds the final error
        res = 2.0 * torch.linalg.matrix_norm(T - eye) / (n ** 0.5)
        res = res * res
        if capturing:
            # a recording cannot hand data to the host: keep the residual in a tensor owned by
            # the graph and let the driver look at it now and then (needs_more_iterations)
            self._graph_residual = res
        elif gpu:
            if self._pending is None:
                # one residual in flight at a time; pinned buffer and event are allocated once
                # (a pinned allocation per call costs more host time than the whole iteration)
                if self._host is None or self._host.numel() != 1:
                    self._host = torch.empty(1, dtype=torch.float32, pin_memory=True)
                    self._event = torch.cuda.Event()
                self._host.copy_(res.reshape(1), non_blocking=True)
                self._event.record()
                self._pending = (self._event, self._host, self.iters)
        else:
            self._pending = (None, res.reshape(1), self.iters)
        rs = torch.sqrt(s)
        sqrtC = 0.5 * (Y + Y.T) * rs
        invsqrtC = 0.5 * (Z + Z.T) / rs
        return sqrtC, invsqrtC
