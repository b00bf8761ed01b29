"""Tiny dense linear systems (m ≤ 16) that sit on per-generation paths.

The library routes (``lstsq``, ``solve``, ``matrix_rank``) cost 0.3–2 ms of host time per call on
a 3 × 3 system; several multi-objective algorithms solve one per generation for a hyperplane
through m extreme points.  On the GPU ``solve_small`` is one single-lane elimination kernel
(``csrc/moproblems.hip``) that also reports rank deficiency, with no host synchronisation."""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, stream_of

_native.register_hip_api({"evox_hip_small_solve": (_native.I, [_native.P, _native.P, _native.I, _native.P, _native.P, _native.P])})


def solve_small(A: torch.Tensor, b: torch.Tensor):
    """``(x, ok)``: the solution of ``A x = b`` for a square ``A`` and whether ``A`` has full rank
    (``ok`` is a 0-d boolean tensor; ``x`` is NaN where it is false on the GPU path)."""
    m = A.shape[0]
    if A.is_cuda and A.dtype == torch.float32 and m <= 16:
        A = A.contiguous()
        b = b.to(torch.float32).contiguous().reshape(m)
        x = torch.empty(m, dtype=torch.float32, device=A.device)
        flag = torch.empty(1, dtype=torch.int32, device=A.device)
        check(hip().evox_hip_small_solve(A.data_ptr(), b.data_ptr(), m, x.data_ptr(), flag.data_ptr(), stream_of(A)),
              "small_solve")
        return x, flag[0] == 0
    x = torch.linalg.lstsq(A, b.reshape(m, 1)).solution[:, 0]
    return x, torch.linalg.matrix_rank(A) == m


# --------------------------------------------------------------------------------------------------------------------
# small dense factorisations on the device (``csrc/linalg_small.hip``): one workgroup per matrix, no host round trip
# --------------------------------------------------------------------------------------------------------------------
import os

P, I = _native.P, _native.I
_native.register_hip_api({
    "evox_hip_small_max_lds_floats": (I, []),
    "evox_hip_small_eigh": (I, [P, P, P, P, P, I, I, I, _native.F, P]),
    "evox_hip_small_eigh_spd": (I, [P, P, P, P, P, P, I, I, I, _native.F, P]),
    "evox_hip_small_eigh_spd_fits": (I, [I, I]),
    "evox_hip_small_svd_vt": (I, [P, P, P, I, I, I, I, _native.F, P]),
    "evox_hip_small_cholesky": (I, [P, P, I, I, P]),
    "evox_hip_small_qr_q": (I, [P, P, I, I, I, P]),
    "evox_hip_small_expm": (I, [P, P, I, I, P]),
})

_DEVICE_MAX_N = 384           # largest order the one-workgroup kernels are meant for
_JACOBI_TOL = 1e-6


def device_linalg(x: torch.Tensor, n: int) -> bool:
    """Whether the factorisation of an order-``n`` fp32 matrix runs in the one-workgroup device kernels
    (``EVOX_AMD_SMALL_LINALG=library`` sends everything to the tensor library instead)."""
    return (x.is_cuda and x.dtype == torch.float32 and 0 < n <= _DEVICE_MAX_N
            and os.environ.get("EVOX_AMD_SMALL_LINALG", "device") == "device")


def eigh_small(C: torch.Tensor, B0: torch.Tensor = None, sort: bool = True, max_sweeps: int = 24, spd: bool = False,
               sweeps_out: torch.Tensor = None):
    """``(evals, B)`` of the symmetric (positive semi-definite) ``C`` — (n, n) or (batch, n, n) — by one-sided Jacobi
    rotations, ``C = B diag(evals) Bᵀ`` with the eigenvectors in the columns of ``B``.  ``B0``: eigenvectors of a nearby
    matrix (the previous generation's): the iteration then starts from ``C B0`` and needs one or two sweeps instead of
    eight.  ``sort``: ascending eigenvalues as ``torch.linalg.eigh`` returns them (tensor ops on the device).

    ``spd=True`` declares ``C`` positive definite (a covariance matrix): the rotations then work on its Cholesky factor
    (of ``B0ᵀ C B0`` with ``B0``), which moves half the data per rotation, converges in 7–8 sweeps at any conditioning
    and in three from a nearby start; orders whose working matrices do not fit the LDS (n > 188, or n > 134 with
    ``B0``) take the general form.  A direction along which ``C`` is singular to working precision (Cholesky pivot
    below 1e-6 of its diagonal entry) comes back with eigenvalue 0 and a zero vector.  ``sweeps_out``: an int32 tensor of
    ``batch`` entries that receives the sweeps taken (``spd`` only)."""
    single = C.dim() == 2
    Cb = (C[None] if single else C).contiguous()
    b, n = Cb.shape[0], Cb.shape[-1]
    V = torch.empty_like(Cb)
    ev = torch.empty((b, n), dtype=torch.float32, device=C.device)
    V0 = None
    if B0 is not None:
        V0 = (B0[None] if single else B0).transpose(-1, -2).contiguous()      # vectors as rows = column-major columns
    if spd and hip().evox_hip_small_eigh_spd_fits(n, int(V0 is not None)):
        scratch = torch.empty_like(Cb) if V0 is not None else None
        check(hip().evox_hip_small_eigh_spd(Cb.data_ptr(), V0.data_ptr() if V0 is not None else None, V.data_ptr(),
                                            ev.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                                            sweeps_out.data_ptr() if sweeps_out is not None else None, n, b,
                                            int(max_sweeps), _JACOBI_TOL, stream_of(Cb)), "small_eigh_spd")
    else:
        W = torch.empty_like(Cb)
        check(hip().evox_hip_small_eigh(Cb.data_ptr(), V0.data_ptr() if V0 is not None else None, V.data_ptr(),
                                        ev.data_ptr(), W.data_ptr(), n, b, int(max_sweeps), _JACOBI_TOL, stream_of(Cb)),
              "small_eigh")
    B = V.transpose(-1, -2)
    if sort:
        order = torch.argsort(ev, dim=-1)
        ev = torch.gather(ev, -1, order)
        B = torch.gather(B, -1, order[:, None, :].expand(b, n, n))
    return (ev[0], B[0]) if single else (ev, B)


def svd_vt_small(X: torch.Tensor, max_sweeps: int = 24):
    """``(Vt, s)``: the right singular vectors (rows) and singular values of ``X`` (k, m), k ≤ m, in descending order,
    by one-sided Jacobi rotations of the rows of ``X``."""
    k, m = X.shape
    work = X.detach().clone().contiguous()
    Vt = torch.empty_like(work)
    s = torch.empty(k, dtype=torch.float32, device=X.device)
    check(hip().evox_hip_small_svd_vt(work.data_ptr(), Vt.data_ptr(), s.data_ptr(), k, m, 1, int(max_sweeps), _JACOBI_TOL,
                                      stream_of(work)), "small_svd_vt")
    return Vt, s


def cholesky_small(A: torch.Tensor) -> torch.Tensor:
    """Lower Cholesky factor of the symmetric positive definite ``A`` ((n, n) or (batch, n, n)); NaNs from the first
    non-positive pivot on."""
    single = A.dim() == 2
    Ab = (A[None] if single else A).contiguous()
    L = torch.empty_like(Ab)
    check(hip().evox_hip_small_cholesky(Ab.data_ptr(), L.data_ptr(), Ab.shape[-1], Ab.shape[0], stream_of(Ab)), "small_cholesky")
    return L[0] if single else L


def qr_q_small(A: torch.Tensor) -> torch.Tensor:
    """``Q`` (m, k) of the thin QR factorisation of ``A`` (m, k), k ≤ m, with the sign convention of a positive
    diagonal of R (Gram–Schmidt with re-orthogonalisation)."""
    m, k = A.shape
    rows = A.T.contiguous()                       # row j = column j
    Q = torch.empty_like(rows)
    check(hip().evox_hip_small_qr_q(rows.data_ptr(), Q.data_ptr(), m, k, 1, stream_of(rows)), "small_qr_q")
    return Q.T


def qr_fits(m: int, k: int) -> bool:
    return k <= m and m + k + 64 <= int(hip().evox_hip_small_max_lds_floats())


def expm_fits(n: int) -> bool:
    return 3 * n * n <= int(hip().evox_hip_small_max_lds_floats())


def expm_small(A: torch.Tensor) -> torch.Tensor:
    """Matrix exponential of ``A`` ((n, n) or (batch, n, n), 3 n² floats within the working storage: n ≤ 110) by scaling
    and squaring with fixed parameters — no norm estimate comes back to the host, the call can be recorded."""
    single = A.dim() == 2
    Ab = (A[None] if single else A).contiguous()
    E = torch.empty_like(Ab)
    check(hip().evox_hip_small_expm(Ab.data_ptr(), E.data_ptr(), Ab.shape[-1], Ab.shape[0], stream_of(Ab)), "small_expm")
    return E[0] if single else E
