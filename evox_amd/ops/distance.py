"""Pairwise Euclidean distances.

On the GPU the distances come from a direct-difference kernel (``csrc/distance.hip``): the
``‖a‖² + ‖b‖² − 2a·b`` form of the tensor library loses small distances to cancellation, and its
direct form returns wrong values for results beyond 2³² / 256 elements on this ROCm build (measured:
errors of order one for a 2926 × 10000 result).  On the host the library's direct form is used."""
from __future__ import annotations

import torch

from .. import _native
from ._util import check, hip, on_gpu, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({"evox_hip_cdist": (I, [P, P, I, I, I, P, I, P])})

_MAX_ROWS = 65535 * 32


def cdist(x: torch.Tensor, y: torch.Tensor, squared: bool = False, p: float = 2.0) -> torch.Tensor:
    """``out[i, j] = ‖x_i − y_j‖_p`` for ``x`` (rows, m) and ``y`` (r, m); ``p`` ∈ {1, 2, inf}; ``squared``
    returns the squared Euclidean distance."""
    if p not in (1.0, 2.0, float("inf")):
        raise ValueError("p must be 1, 2 or inf")
    if squared and p != 2.0:
        raise ValueError("squared distances exist for p = 2 only")
    mode = 1 if squared else {2.0: 0, 1.0: 2, float("inf"): 3}[float(p)]
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise ValueError("cdist expects (p, m) and (r, m)")
    if on_gpu(x) and x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape[1] > 0 and x.shape[0] <= _MAX_ROWS:
        n, m = x.shape
        r = y.shape[0]
        out = torch.empty((n, r), dtype=torch.float32, device=x.device)
        if n and r:
            xc, yc = x.contiguous(), y.contiguous()
            check(hip().evox_hip_cdist(xc.data_ptr(), yc.data_ptr(), n, r, m, out.data_ptr(), mode, stream_of(x)), "cdist")
        return out
    if p != 2.0:
        if x.is_cuda:                                      # broadcast form: the library's direct kernel is the faulty one
            diff = (x[:, None, :] - y[None, :, :]).abs()
            return diff.sum(-1) if p == 1.0 else diff.amax(-1)
        return torch.cdist(x, y, p=p)
    if x.is_cuda:
        d = torch.cdist(x, y)                              # GEMM form: correct at every size, approximate near zero
    else:
        d = torch.cdist(x, y, compute_mode="donot_use_mm_for_euclid_dist")
    return d.square() if squared else d
