"""GEMM-shaped hot ops on the gfx950 matrix cores (kernels: ``csrc/mfma_gemm.hip``).

``matmul_nt / matmul``      fp32 GEMM with exact-fp32 MFMA (plain library-style GEMM).
``cma_sample``              X = mean + σ (Z∘d) Bᵀ with Z drawn inside the kernel.
``cma_covariance_update``   C' = αC + c₁ pc pcᵀ + cμ Yᵀ diag(w) Y with Y gathered on the fly.
"""
from __future__ import annotations

import os

import torch

from .. import _native
from ..core import random as rnd
from . import blas as _blas
from ._util import check, hip, stream_of

P, I, I64, F = _native.P, _native.I, _native.I64, _native.F
_native.register_hip_api({
    "evox_hip_gemm_f32": (I, [P, P, P, P, I, I, I, I64, I64, I64, I64, I, I, F, F, P]),
    "evox_hip_cma_sample": (I, [P, P, P, P, P, P, I, I, I64, P]),
    "evox_hip_cma_cov": (I, [P, P, P, P, P, P, P, P, F, F, P, I, I, I64, P, I, P]),
    "evox_hip_cma_gather_scale": (I, [P, I64, P, P, P, P, P, I, P, I, I, I, P]),
    "evox_hip_cma_affine": (I, [P, P, P, I64, I, P]),
    "evox_hip_cma_paths": (I, [P, P, P, P, P, P, P, P, I, P, P, P, P, P]),
    "evox_hip_cma_cov_base": (I, [P, P, P, F, P, I, P]),
})


def _impl() -> str:
    return os.environ.get("EVOX_AMD_GEMM", "hip")


def _layout(x: torch.Tensor):
    """(tensor, leading dimension, k_major) for a 2-D operand used as rows × K.

    A row-major contiguous (R, K) tensor is K-contiguous; its transpose view (strides (1, ld))
    is k-major.  Anything else is made contiguous."""
    if x.stride(1) == 1 and x.stride(0) >= x.shape[1]:
        return x, x.stride(0), 0
    if x.stride(0) == 1 and x.stride(1) >= x.shape[0]:
        return x, x.stride(1), 1
    x = x.contiguous()
    return x, x.stride(0), 0


def matmul_nt(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, beta: float = 0.0, c0=None, out=None):
    """``alpha * a @ b.T + beta * c0`` for fp32 ``a`` (M, K) and ``b`` (N, K) in any stride order."""
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError("matmul_nt expects fp32 operands")
    M, K = a.shape
    N, K2 = b.shape
    if K != K2:
        raise ValueError(f"inner dimensions differ: {K} vs {K2}")
    a, lda, ak = _layout(a)
    b, ldb, bk = _layout(b)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    elif not out.is_contiguous() or tuple(out.shape) != (M, N):
        raise ValueError("out must be a contiguous (M, N) tensor")
    if c0 is not None:
        c0 = c0.contiguous()
        if tuple(c0.shape) != (M, N):
            raise ValueError("c0 must have shape (M, N)")
    check(hip().evox_hip_gemm_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), c0.data_ptr() if c0 is not None else None,
                                  M, N, K, lda, ldb, N, N, ak, bk, float(alpha), float(beta), stream_of(a)), "gemm_f32")
    return out


def matmul(a: torch.Tensor, b: torch.Tensor, **kw):
    """``a @ b`` (fp32, exact-fp32 MFMA)."""
    return matmul_nt(a, b.T, **kw)


_epilogue_ok = [True]


def _sample_impl() -> str:
    """How the sampling GEMM runs on the GPU.

    ``library``  Z is drawn by the Philox kernel into a (λ, n) buffer, the product Z·(B∘d)ᵀ is a plain
                 library GEMM and mean/σ are applied by one in-place pass.  Fastest measured form.
    ``fused``    the hand-written MFMA kernel that draws Z inside its prologue (no Z buffer); it
                 re-draws each Z tile once per column block of the output, which costs more than
                 the buffer saves at n = 10³.
    ``mfma``     Z buffer + the hand-written exact-fp32 MFMA GEMM."""
    return os.environ.get("EVOX_AMD_CMA_SAMPLE", "library")


def cma_sample(key, mean, sigma, B, D, pop_size, precision="fp32", row_offset=0):
    n = mean.shape[0]
    impl = "torch" if _impl() == "torch" else _sample_impl()
    if impl == "torch":
        z = rnd.normal(key, (pop_size, n), offset=row_offset * n)
        zd = z if D is None else z * D
        if precision == "bf16":
            return mean + sigma * (zd.bfloat16() @ B.T.bfloat16()).float()
        return mean + sigma * (zd @ B.T)
    if impl != "fused" or precision == "bf16":
        z = rnd.normal(key, (pop_size, n), offset=row_offset * n,
                       dtype=torch.bfloat16 if precision == "bf16" else torch.float32)
        bd = B if D is None else B * D                            # (n, n): column j of B scaled by d_j
        sig = torch.as_tensor(sigma, dtype=torch.float32, device=mean.device).reshape(1)
        vec = getattr(sigma, "_evox_vec", None)         # σ already expanded by the path kernel
        if vec is not None and vec.numel() == n and vec.data_ptr() % 16 == 0 and vec.device == mean.device:
            sig_vec = vec
        else:
            sig_vec = sig
        if impl != "mfma" and _blas.available() and _epilogue_ok[0]:
            # mean + σ·Z·(B∘d)ᵀ from the GEMM alone: σ is read on the device, the mean is the
            # epilogue's bias vector
            try:
                if precision == "bf16":
                    # the factorisation may have left a rounded copy of B next to it
                    b16 = getattr(B, "_evox_bf16", None) if D is None else None
                    return _blas.gemm(z.bfloat16(), b16 if b16 is not None else bd.bfloat16(), trans_b=True, alpha=sig_vec,
                                      bias=mean)
                return _blas.gemm(z, bd, trans_b=True, alpha=sig_vec, bias=mean)
            except RuntimeError as e:
                # a hipBLASLt build without this epilogue / scale mode: say so once, use the plain product
                import warnings
                warnings.warn(f"library GEMM with fused scale and bias is unavailable ({e}); using GEMM + affine pass")
                _epilogue_ok[0] = False
        if precision == "bf16":
            g = torch.mm(z.bfloat16(), bd.bfloat16().T, out_dtype=torch.float32)
        elif impl == "mfma":
            g = matmul_nt(z, bd)
        else:
            g = torch.mm(z, bd.T)
        check(hip().evox_hip_cma_affine(g.data_ptr(), mean.contiguous().data_ptr(), sig.data_ptr(), pop_size, n,
                                        stream_of(mean)), "cma_affine")
        return g
    if D is None:
        D = torch.ones(n, dtype=torch.float32, device=mean.device)
    if tuple(B.shape) != (n, n) or D.shape[0] != n:
        raise ValueError("B must be (n, n) and D (n,)")
    B = B.contiguous()
    sigma = torch.as_tensor(sigma, dtype=torch.float32, device=mean.device).reshape(1)
    out = torch.empty((pop_size, n), dtype=torch.float32, device=mean.device)
    check(hip().evox_hip_cma_sample(key.contiguous().data_ptr(), mean.contiguous().data_ptr(), sigma.data_ptr(),
                                    B.data_ptr(), D.contiguous().data_ptr(), out.data_ptr(), pop_size, n,
                                    int(row_offset), stream_of(mean)), "cma_sample")
    return out


def cma_elite_factor(pop, order, mean, sigma, sqrt_w, precision="fp32"):
    """``(S, step)``: ``S[i] = sqrt(w_i) (x_{order[i]} − mean) / σ`` for the μ selected rows (fp32,
    or bf16 when ``precision == "bf16"``) and the weighted mean shift ``step = Σ w_i (x_i − mean)``
    (always accumulated in fp32 from the unrounded differences)."""
    mu, n = sqrt_w.shape[0], mean.shape[0]
    if pop.stride(1) != 1:
        pop = pop.contiguous()
    if order.dtype != torch.int64:
        order = order.to(torch.int64)
    dev = pop.device
    sig = torch.as_tensor(sigma, dtype=torch.float32, device=dev).reshape(1)
    bf16 = precision == "bf16"
    out = torch.empty((mu, n), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
    blocks = max(1, min(mu, int(os.environ.get("EVOX_AMD_GATHER_BLOCKS", "512"))))
    colsum = torch.empty((blocks, n), dtype=torch.float32, device=dev)
    check(hip().evox_hip_cma_gather_scale(pop.data_ptr(), pop.stride(0), order.contiguous().data_ptr(),
                                          mean.contiguous().data_ptr(), sqrt_w.contiguous().data_ptr(), sig.data_ptr(),
                                          out.data_ptr(), 1 if bf16 else 0, colsum.data_ptr(), blocks, mu, n,
                                          stream_of(pop)), "cma_gather_scale")
    return out, colsum.sum(0)


def cma_paths(mean, step, whitened, ps, pc, sigma, iter_dev, consts):
    """Fused path / step-size update (``csrc/cma_aux.hip``).  ``whitened = C^{-1/2} step``;
    ``consts`` = (cs, cc, mueff, cm, damps, chiN, c1, cmu) as a host float array.
    Returns ``(mean', ps', pc', sigma', alpha, hsig)`` — the last three as 0-d views."""
    import ctypes

    n = mean.shape[0]
    dev = mean.device
    mean_out, ps_out, pc_out = torch.empty_like(mean), torch.empty_like(ps), torch.empty_like(pc)
    scalars = torch.empty(4 + n, dtype=torch.float32, device=dev)         # 4 scalars, then sigma' repeated n times
    c_arr = (ctypes.c_float * 8)(*[float(v) for v in consts])
    check(hip().evox_hip_cma_paths(mean.contiguous().data_ptr(), step.contiguous().data_ptr(),
                                   whitened.contiguous().data_ptr(), ps.contiguous().data_ptr(), pc.contiguous().data_ptr(),
                                   sigma.reshape(1).data_ptr(), iter_dev.reshape(1).data_ptr(), ctypes.addressof(c_arr), n,
                                   mean_out.data_ptr(), ps_out.data_ptr(), pc_out.data_ptr(), scalars.data_ptr(),
                                   stream_of(mean)), "cma_paths")
    sigma_new = scalars[0]
    sigma_new._evox_vec = scalars[4:]                  # consumed by cma_sample (scale vector of the GEMM epilogue)
    return mean_out, ps_out, pc_out, sigma_new, scalars[1], scalars[2]


def cma_covariance_base(C, pc, alpha, c1):
    """``αC + c₁ pc pcᵀ`` in one pass (α may be a device scalar: it depends on the stall indicator)."""
    n = C.shape[0]
    if isinstance(alpha, torch.Tensor) and C.dtype == torch.float32 and C.is_cuda:
        base = torch.empty_like(C, memory_format=torch.contiguous_format)
        check(hip().evox_hip_cma_cov_base(C.contiguous().data_ptr(), pc.contiguous().data_ptr(),
                                          alpha.reshape(1).data_ptr(), float(c1), base.data_ptr(), n, stream_of(C)),
              "cma_cov_base")
        return base
    return torch.addr(C * alpha, pc, pc, alpha=float(c1))


def gram(S, out=None):
    """``SᵀS`` in fp32 (``S`` fp32 or bf16), optionally into a given row-major fp32 matrix."""
    if S.is_cuda and _blas.available() and S.data_ptr() % 16 == 0 and (out is None or out.data_ptr() % 16 == 0):
        return _blas.gemm(S, S, trans_a=True, out=out)
    g = torch.mm(S.T, S, out_dtype=torch.float32) if S.dtype == torch.bfloat16 else torch.mm(S.T, S)
    if out is not None:
        out.copy_(g)
        return out
    return g


def cma_covariance_from_factor(C, S, pc, alpha, c1, cmu):
    """``αC + c₁ pc pcᵀ + cμ SᵀS``: one pass forms the first two terms, the library GEMM
    accumulates the Gram matrix onto them."""
    base = cma_covariance_base(C, pc, alpha, c1)
    if _blas.available():
        return _blas.gemm(S, S, trans_a=True, alpha=float(cmu), beta=1.0, out=base)
    if S.dtype == torch.bfloat16:
        return base + float(cmu) * torch.mm(S.T, S, out_dtype=torch.float32)
    return torch.addmm(base, S.T, S, alpha=float(cmu))


def cma_covariance_update(C, pop, order, mean, sigma, weights, pc, alpha, c1, cmu, precision="fp32"):
    n = C.shape[0]
    mu = weights.shape[0]
    if _impl() == "torch" or precision == "bf16":
        y = (pop[order[:mu]] - mean) / sigma
        if precision == "bf16":
            g = ((y.T * weights).bfloat16() @ y.bfloat16()).float()
        else:
            g = (y.T * weights) @ y
        return alpha * C + c1 * torch.outer(pc, pc) + cmu * g
    if order.dtype != torch.int64:
        order = order.to(torch.int64)
    if order.shape[0] < mu:
        raise ValueError("order is shorter than the number of recombination weights")
    if pop.stride(1) != 1:
        pop = pop.contiguous()
    dev = C.device
    sigma = torch.as_tensor(sigma, dtype=torch.float32, device=dev).reshape(1)
    alpha = torch.as_tensor(alpha, dtype=torch.float32, device=dev).reshape(1)
    out = torch.empty_like(C, memory_format=torch.contiguous_format)
    # enough K-slices that (upper-triangular tiles x slices) covers the 256 CUs about twice
    tiles = ((n + 63) // 64) * ((n + 63) // 64 + 1) // 2
    splits = max(1, min(8, 512 // max(tiles, 1), mu // 256))
    ws = torch.empty((splits, n, n), dtype=torch.float32, device=dev) if splits > 1 else None
    check(hip().evox_hip_cma_cov(C.contiguous().data_ptr(), pop.data_ptr(), order.contiguous().data_ptr(),
                                 mean.contiguous().data_ptr(), sigma.data_ptr(), weights.contiguous().data_ptr(),
                                 pc.contiguous().data_ptr(), alpha.data_ptr(), float(c1), float(cmu), out.data_ptr(),
                                 n, mu, pop.stride(0), ws.data_ptr() if ws is not None else None, splits,
                                 stream_of(C)), "cma_cov")
    return out
