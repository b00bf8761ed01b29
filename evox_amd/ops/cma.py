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
    return mean + sigma * ((z * D) @ B.T)


def sample(key, mean, sigma, B, D, pop_size: int, precision: str = "fp32", row_offset: int = 0):
    """``pop_size`` rows starting at logical row ``row_offset`` of the generation's sample."""
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
def eigh_reference(C) -> Tuple[torch.Tensor, torch.Tensor]:
    C = torch.triu(C) + torch.triu(C, 1).T
    evals, B = torch.linalg.eigh(C)
    return B, evals


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

    def __init__(self, tol: float = 2e-6, min_iters: int = 2, max_iters: int = 40, start_iters: int = 6):
        self.tol, self.min_iters, self.max_iters = tol, min_iters, max_iters
        self.iters = start_iters
        self._pending = None          # (event, pinned residual, iters used)
        self.last_residual = None
        self._graph_residual = None

    def _poll(self):
        if self._pending is None:
            return
        ev, host, used = self._pending
        if ev is not None and not ev.query():
            return
        self._pending = None
        r = float(host[0])
        self.last_residual = r
        if not (r == r) or r > self.tol:
            self.iters = min(self.max_iters, max(self.iters, used) + (4 if (r != r or r > 1e-2) else 2))
        elif r < self.tol * 1e-3 and used <= self.iters:
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

    def __call__(self, C: torch.Tensor):
        n = C.shape[0]
        gpu = C.is_cuda
        capturing = gpu and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self._poll()
        eye = torch.eye(n, device=C.device, dtype=C.dtype)
        # scale towards the centre of the spectrum when that is provably safe (λmax < 2s needs
        # ‖C‖∞ < 2s), else by the norm bound: a centred spectrum saves iterations
        s = torch.maximum(torch.diagonal(C).mean(), C.abs().sum(1).max() / 1.9)
        Y = C / s
        Z = eye
        T = eye
        # the multiplication ORDER below is the numerically stable coupled form; products are
        # plain dense GEMMs and go to the vendor library (rocBLAS/hipBLASLt through torch)
        for _ in range(self.iters):
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
            host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host.copy_(res.reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(C.device))
            self._pending = (ev, host, self.iters)
        else:
            self._pending = (None, res.reshape(1), self.iters)
        rs = torch.sqrt(s)
        sqrtC = 0.5 * (Y + Y.T) * rs
        invsqrtC = 0.5 * (Z + Z.T) / rs
        return sqrtC, invsqrtC
