"""Fused differential-evolution variation (``csrc/de.hip``): picks + mutation + binomial crossover +
bounds in one launch, drawing the same Philox elements as the batched PyTorch formulation in
``algorithms/so/de_variants/base.py`` (which stays the host path and the oracle)."""
from __future__ import annotations

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, stream_of

P, I, I64 = _native.P, _native.I, _native.I64
_native.register_hip_api({
    "evox_hip_de_trial": (I, [P, I64, I, I, P, I, P, P, P, I, P, I, I, I, I, I, P, P, P, P, P, P, P]),
})


def _vec(x, b, dev):
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        return t, int(t.numel() == b and b > 1) if t.numel() != 1 else 0
    return torch.full((1,), float(x), dtype=torch.float32, device=dev), 0


def supported(pop, prim, sec, num_diff, cross, n_picks, replace) -> bool:
    return (pop.is_cuda and pop.dtype == torch.float32 and isinstance(prim, int) and isinstance(sec, int)
            and isinstance(num_diff, int) and isinstance(cross, int) and cross == 0 and not replace
            and 1 <= num_diff <= 2 and 2 * num_diff + 1 <= n_picks <= 9 and pop.shape[0] > n_picks)


def de_trial(key, pop, idx, best_index, pbest_rows, prim, sec, num_diff, F, CR, n_picks, lb, ub):
    """Trial vectors for the rows ``idx`` of ``pop``; ``key`` is split exactly as the PyTorch path does
    (selection / pbest / crossover → jrand, mask)."""
    dev = pop.device
    b, dim = idx.shape[0], pop.shape[1]
    k_sel, _k_pb, k_x = rnd.split(key, 3)
    k_r, k_m = rnd.split(k_x)
    if pop.stride(1) != 1:
        pop = pop.contiguous()
    f, f_vec = _vec(F, b, dev)
    cr, cr_vec = _vec(CR, b, dev)
    out = torch.empty((b, dim), dtype=torch.float32, device=dev)
    check(hip().evox_hip_de_trial(
        pop.data_ptr(), pop.stride(0), pop.shape[0], dim, idx.contiguous().data_ptr(), b,
        best_index.reshape(1).contiguous().data_ptr(), pbest_rows.contiguous().data_ptr() if pbest_rows is not None else None,
        f.data_ptr(), f_vec, cr.data_ptr(), cr_vec, int(prim), int(sec), int(num_diff), int(n_picks),
        k_sel.contiguous().data_ptr(), k_r.contiguous().data_ptr(), k_m.contiguous().data_ptr(),
        lb.contiguous().data_ptr(), ub.contiguous().data_ptr(), out.data_ptr(), stream_of(pop)), "de_trial")
    return out
