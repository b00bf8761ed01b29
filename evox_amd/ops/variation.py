"""Fused variation operators (``csrc/variation.hip``).

``sbx_pm``  simulated-binary crossover → polynomial mutation → bound clamp in one launch, drawing
            the same Philox elements as ``operators.crossover.simulated_binary`` followed by
            ``operators.mutation.polynomial`` (reference: ``src/evox/operators/crossover/sbx.py:7-40``,
            ``src/evox/operators/mutation/pm_mutation.py:6-54``)."""
from __future__ import annotations

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of

P, I, I64, F = _native.P, _native.I, _native.I64, _native.F
_native.register_hip_api({"evox_hip_sbx_pm": (I, [P, I64, I, I, P, P, P, F, F, F, F, P, P])})


def sbx_pm_supported(x: torch.Tensor, lb: torch.Tensor, ub: torch.Tensor) -> bool:
    return (on_gpu(x) and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and lb.dtype == torch.float32
            and lb.numel() == x.shape[1] and ub.numel() == x.shape[1])


def sbx_pm(key_x, key_m, x, lb, ub, pro_c=1.0, dis_c=20.0, pro_m=1.0, dis_m=20.0):
    """Offspring of the parent matrix ``x`` (first half mated with second half), mutated and clamped."""
    n, d = x.shape
    h = n // 2
    if x.stride(1) != 1:
        x = x.contiguous()
    keys = torch.cat([rnd.split(key_x, 4), rnd.split(key_m, 2)], dim=0).contiguous()
    out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    check(hip().evox_hip_sbx_pm(x.data_ptr(), x.stride(0), h, d, keys.data_ptr(), lb.contiguous().data_ptr(),
                                ub.contiguous().data_ptr(), float(pro_c), float(dis_c), float(pro_m), float(dis_m),
                                out.data_ptr(), stream_of(x)), "sbx_pm")
    if n % 2:
        out[-1] = torch.clamp(x[-1], lb, ub)
    return out
