"""Library GEMMs issued straight from the native runtime (``csrc/blas.hip``).

``gemm``      ``alpha * op(a) @ op(b) + beta * c`` through a cached hipBLASLt plan whose algorithm
              was chosen by timing the heuristic's candidates once on the device.  One call costs a
              few µs of host time instead of the 13–26 µs of the tensor library's matmul path,
              which is what bounds a generation made of ~30 mid-sized products.
``ns_sqrt``   the whole coupled Newton–Schulz iteration (scaling by a power-iteration estimate of
              λmax, the GEMM chain, residual and symmetrised outputs) from one native call.

The hipBLASLt instance is the one the process already holds (the tensor library's copy); it is
opened by path, never linked, so two ROCm releases are not mixed in one process.
"""
from __future__ import annotations

import ctypes
import os

import torch

from .. import _native
from ._util import check, hip, stream_of

P, I, I64, F = _native.P, _native.I, _native.I64, _native.F
_native.register_hip_api({
    "evox_hip_blas_init": (I, [ctypes.c_char_p]),
    "evox_hip_gemm": (I, [I, I, I, I64, I64, I64, F, P, I64, P, I64, F, P, I64, P, I64, P, P, I, P]),
    "evox_hip_gemm_plan_info": (I, [I, I, I, I64, I64, I64, I64, I64, I64, I64, I, P]),
    "evox_hip_ns_scal_floats": (I, []),
    "evox_hip_ns_sqrt": (I, [P, I, I, I, I, P, P, P, P, P, P, P, P, I, I, P, P]),
    "evox_hip_ns_cond": (I, [P, I, P, P, P, P]),
})

_ready = None


def _candidates():
    env = os.environ.get("EVOX_AMD_HIPBLASLT")
    if env:
        yield env
    yield os.path.join(os.path.dirname(torch.__file__), "lib", "libhipblaslt.so")
    yield "libhipblaslt.so.1"
    yield "/opt/rocm/lib/libhipblaslt.so"


def available() -> bool:
    """True once the hipBLASLt entry points are resolved (GPU build only)."""
    global _ready
    if _ready is None:
        _ready = False
        if torch.cuda.is_available() and os.environ.get("EVOX_AMD_BLAS", "native") == "native":
            lib = hip()
            for path in _candidates():
                if (os.path.isabs(path) and not os.path.exists(path)):
                    continue
                if lib.evox_hip_blas_init(path.encode()) == 0:
                    _ready = True
                    break
    return _ready


def _tune() -> int:
    return 0 if os.environ.get("EVOX_AMD_BLAS_TUNE", "1") == "0" else 1


def _row_major(x: torch.Tensor) -> torch.Tensor:
    return x if (x.stride(1) == 1 and x.stride(0) >= x.shape[1]) else x.contiguous()


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
         beta: float = 0.0, c: torch.Tensor | None = None, out: torch.Tensor | None = None,
         bias: torch.Tensor | None = None) -> torch.Tensor:
    """``alpha * op(a) @ op(b) + beta * c + bias`` with fp32 accumulation and output.

    ``a`` and ``b`` are 2-D row-major tensors of the same dtype (fp32 or bf16); ``trans_*`` select
    the transposed use without a copy.  ``c`` (fp32) may be a different tensor from ``out``.
    ``alpha`` may be an fp32 device tensor, one value or one per output column (read on the device).
    ``bias`` is an fp32 vector of length N added to every row in the GEMM's epilogue."""
    if not available():
        raise RuntimeError("hipBLASLt could not be opened")
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("gemm expects two fp32 or two bf16 operands")
    a, b = _row_major(a), _row_major(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    K2, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != K2:
        raise ValueError(f"inner dimensions differ: {K} vs {K2}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (M, N) or out.stride(1) != 1:
        raise ValueError("out must be a row-major fp32 (M, N) tensor")
    if beta != 0.0:
        if c is None:
            c = out
        elif c.dtype != torch.float32 or tuple(c.shape) != (M, N) or c.stride(1) != 1:
            raise ValueError("c must be a row-major fp32 (M, N) tensor")
    else:
        c = None
    if any(t is not None and t.data_ptr() % 16 for t in (a, b, c, out)):
        # plans are cached per shape and were timed on 16-byte aligned operands: a view that starts
        # elsewhere could be handed to a kernel that assumes vector-aligned rows
        raise ValueError("gemm operands must start on a 16-byte boundary")
    alpha_dev = None
    if isinstance(alpha, torch.Tensor):
        if alpha.dtype != torch.float32 or alpha.numel() not in (1, N) or alpha.device != a.device:
            raise ValueError("a tensor alpha must hold one fp32 value, or one per output column, on the operands' device")
        alpha_dev = (alpha.reshape(1).expand(N) if alpha.numel() == 1 else alpha.reshape(N)).contiguous()
        alpha = 1.0
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N or bias.device != a.device:
            raise ValueError("bias must be an fp32 vector of length N on the operands' device")
        bias = bias.contiguous()
    check(hip().evox_hip_gemm(1 if a.dtype == torch.bfloat16 else 0, int(trans_a), int(trans_b), M, N, K, float(alpha),
                              a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), float(beta),
                              c.data_ptr() if c is not None else None, c.stride(0) if c is not None else 0,
                              out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None,
                              alpha_dev.data_ptr() if alpha_dev is not None else None, _tune(), stream_of(a)),
          "hipblaslt gemm")
    return out


def plan_info(a_dtype, trans_a, trans_b, M, N, K, lda, ldb, ldc, ldd, beta_nonzero=False, c_is_not_d=False,
              has_bias=False, alpha_on_device=False):
    """(number of candidate algorithms, best timed µs) of a plan that has been used already."""
    info = torch.zeros(2, dtype=torch.float32)
    flags = int(beta_nonzero) | int(c_is_not_d) << 1 | int(has_bias) << 2 | int(alpha_on_device) << 3
    rc = hip().evox_hip_gemm_plan_info(1 if a_dtype == torch.bfloat16 else 0, int(trans_a), int(trans_b), M, N, K,
                                       lda, ldb, ldc, ldd, flags, info.data_ptr())
    return None if rc else (int(info[0]), float(info[1]))


class NSWorkspace:
    """Device scratch of the native Newton–Schulz driver for one matrix size: six n×n work
    matrices, the identity, the power-iteration vector carried between calls and the scalar /
    partial-result block."""

    def __init__(self, n: int, device):
        self.n = n
        self.work = torch.empty((6, n, n), dtype=torch.float32, device=device)
        self.eye = torch.eye(n, dtype=torch.float32, device=device)
        g = torch.Generator().manual_seed(0x5EED)
        v = torch.randn(n, generator=g)
        self.vec = torch.zeros(2 * n, dtype=torch.float32, device=device)
        self.vec[:n] = (v / v.norm()).to(device)
        self.scal = torch.zeros(hip().evox_hip_ns_scal_floats(), dtype=torch.float32, device=device)
        self.fresh = True


def _first_step_default() -> str:
    return os.environ.get("EVOX_AMD_NS_FIRST", "poly4")


def ns_sqrt(C: torch.Tensor, ws: NSWorkspace, iters: int, *, power_steps: int = 1, safe: bool = False,
            prev: tuple | None = None, first_step: str | None = None, bf16_copy: bool = False):
    """``(sqrtC, invsqrtC)`` of the SPD matrix ``C`` after ``iters`` coupled Newton–Schulz steps.

    ``first_step``: ``"poly4"`` (default, or ``EVOX_AMD_NS_FIRST``) makes the first of them the fifth-order
    step built on the degree-4 Taylor polynomial of the inverse square root (3 products instead of 1, worth two
    ordinary steps); ``"newton"`` keeps every step quadratic.  ``bf16_copy``: the final pass also writes the
    square root rounded to bf16 and attaches it to the returned fp32 tensor as ``_evox_bf16`` (consumers that
    feed it to a bf16 GEMM skip their own conversion pass).

    ``ws.scal[13:16]`` afterwards holds the scale used, the residual bound and a flag telling that
    the iteration failed to contract and ``prev`` (the previous factors) was passed through."""
    n = C.shape[0]
    if C.dtype != torch.float32 or tuple(C.shape) != (n, n) or ws.n != n:
        raise ValueError("ns_sqrt expects a square fp32 matrix matching the workspace")
    if not available():
        raise RuntimeError("hipBLASLt could not be opened")
    C = C.contiguous()
    out_s = torch.empty_like(C)
    out_i = torch.empty_like(C)
    if ws.fresh:
        power_steps = 8          # first call: let the power vector settle
        ws.fresh = False
    ps, pi = (prev[0].contiguous(), prev[1].contiguous()) if prev is not None else (None, None)
    out_b = torch.empty((n, n), dtype=torch.bfloat16, device=C.device) if bf16_copy else None
    check(hip().evox_hip_ns_sqrt(C.data_ptr(), n, int(iters), int(power_steps), int(bool(safe)), ws.work.data_ptr(),
                                 ws.eye.data_ptr(), ws.vec.data_ptr(), ws.scal.data_ptr(),
                                 ps.data_ptr() if ps is not None else None, pi.data_ptr() if pi is not None else None,
                                 out_s.data_ptr(), out_i.data_ptr(), _tune(),
                                 int((first_step or _first_step_default()) == "poly4"),
                                 out_b.data_ptr() if out_b is not None else None, stream_of(C)), "ns_sqrt")
    if out_b is not None:
        out_s._evox_bf16 = out_b
    return out_s, out_i


def ns_cond(inv: torch.Tensor, vec2: torch.Tensor, scal: torch.Tensor, scal2: torch.Tensor):
    """Condition estimates of the matrix whose inverse square root is ``inv``, left in ``scal[9]`` (estimate from
    below: λmax·‖inv·v‖², ``v`` = ``vec2[:n]`` turned one power step further) and ``scal[10]`` (bound from above:
    λmax·‖inv‖∞²); ``scal[13]`` must hold the λmax scale of the factorisation that produced ``inv``."""
    n = inv.shape[0]
    check(hip().evox_hip_ns_cond(inv.data_ptr(), n, vec2.data_ptr(), scal.data_ptr(), scal2.data_ptr(), stream_of(inv)),
          "ns_cond")
