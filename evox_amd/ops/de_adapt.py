"""Fused bookkeeping of the self-adaptive DE variants (``csrc/de_adapt.hip``): parameter draws
from the success memory, greedy replacement with success statistics, and the memory update —
three launches per generation (reference formulation: ``src/evox/algorithms/so/de_variants/
shade.py:83-209``, ``jade.py:86-184``)."""
from __future__ import annotations

import os

import torch

from .. import _native
from ._util import check, hip, on_gpu, stream_of

P, I, F = _native.P, _native.I, _native.F
_native.register_hip_api({
    "evox_hip_shade_params": (I, [P, P, P, P, P, I, I, P, P, P]),
    "evox_hip_de_select_blocks": (I, [I, I]),
    "evox_hip_de_select": (I, [P, P, P, P, P, P, I, I, I, I, P, P, P, P]),
    "evox_hip_de_finalize": (I, [P, I, I, F, P, P, I, P, P, P, P]),
})


def supported(pop: torch.Tensor) -> bool:
    return on_gpu(pop) and pop.dtype == torch.float32 and os.environ.get("EVOX_AMD_DE_ADAPT", "hip") == "hip"


def draw_params(k_slot, k_f, k_cr, mem_f, mem_cr, n):
    """``(F, CR)`` for ``n`` individuals: F ~ clip(N(M_F[r], 0.1)), CR ~ clip(Cauchy(M_CR[r], 0.1)) with
    the memory slot ``r`` drawn per individual (one slot: JADE's running means)."""
    mem_f, mem_cr = mem_f.reshape(-1).contiguous(), mem_cr.reshape(-1).contiguous()
    H = mem_f.shape[0]
    Fv = torch.empty(n, dtype=torch.float32, device=mem_f.device)
    CRv = torch.empty(n, dtype=torch.float32, device=mem_f.device)
    check(hip().evox_hip_shade_params(k_slot.contiguous().data_ptr(), k_f.contiguous().data_ptr(),
                                      k_cr.contiguous().data_ptr(), mem_f.data_ptr(), mem_cr.data_ptr(), H, n,
                                      Fv.data_ptr(), CRv.data_ptr(), stream_of(mem_f)), "shade_params")
    return Fv, CRv


def select_and_adapt(pop, trial, fit, trial_fit, Fv, CRv, mem_f, mem_cr, *, strict: bool, weight_by_gain: bool,
                     mode: str, c: float = 0.0):
    """Greedy replacement over the whole population and the adaptation that follows it.

    Returns ``(population, fitness, best_index, new_mem_f, new_mem_cr, packed)`` (``packed`` = both rows).  ``mode="shade"``: the
    memories are the H-slot histories (new means in slot 0, rest shifted); ``mode="jade"``: they are
    the two running means, moved by rate ``c``."""
    n, d = pop.shape
    lib = hip()
    s = stream_of(pop)
    pop, trial = pop.contiguous(), trial.contiguous()
    new_pop = torch.empty_like(pop)
    new_fit = torch.empty_like(fit)
    blocks = lib.evox_hip_de_select_blocks(n, d)
    partials = torch.empty((blocks, 6), dtype=torch.float32, device=pop.device)
    check(lib.evox_hip_de_select(pop.data_ptr(), trial.data_ptr(), fit.contiguous().data_ptr(),
                                 trial_fit.contiguous().data_ptr(), Fv.contiguous().data_ptr(),
                                 CRv.contiguous().data_ptr(), int(strict), int(weight_by_gain), n, d,
                                 new_pop.data_ptr(), new_fit.data_ptr(), partials.data_ptr(), s), "de_select")
    mem_f_c, mem_cr_c = mem_f.reshape(-1).contiguous(), mem_cr.reshape(-1).contiguous()
    H = mem_f_c.shape[0]
    packed = torch.empty((2, H), dtype=torch.float32, device=pop.device)          # row 0: F side, row 1: CR side
    best = torch.empty((), dtype=torch.int64, device=pop.device)
    check(lib.evox_hip_de_finalize(partials.data_ptr(), blocks, 0 if mode == "shade" else 1, float(c), mem_f_c.data_ptr(),
                                   mem_cr_c.data_ptr(), H, packed[0].data_ptr(), packed[1].data_ptr(), best.data_ptr(), s),
          "de_finalize")
    return new_pop, new_fit, best, packed[0].reshape(mem_f.shape), packed[1].reshape(mem_cr.shape), packed
