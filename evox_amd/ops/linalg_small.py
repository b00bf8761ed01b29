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
