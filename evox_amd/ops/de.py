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
    "evox_hip_de_trial": (I, [P, I64, I, I, P, I, P, P, P, I, P, I, I, I, I, I, P, I, P, P, P, P, P, P, I, I, P]),
})


def _vec(x, b, dev):
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        return t, int(t.numel() == b and b > 1) if t.numel() != 1 else 0
    return torch.full((1,), float(x), dtype=torch.float32, device=dev), 0


BIN, EXP, ARITH = 0, 1, 2


def supported(pop, prim, sec, num_diff, cross, n_picks, replace, code_sets=None) -> bool:
    """Static strategies (ints) with binomial crossover or arithmetic recombination, or per-row
    strategy tensors whose possible values the caller declares in ``code_sets`` (``{"cross": {...},
    "num_diff": {...}}``): exponential crossover stays on the tensor path."""
    if not (pop.is_cuda and pop.dtype == torch.float32 and not replace and n_picks <= 9 and pop.shape[0] > n_picks):
        return False
    static = all(isinstance(v, int) for v in (prim, sec, num_diff, cross))
    if static:
        return cross in (BIN, ARITH) and 1 <= num_diff <= 2 and 2 * num_diff + 1 <= n_picks
    if code_sets is None or any(isinstance(v, int) for v in (prim, sec, num_diff, cross)):
        return False
    return (set(code_sets.get("cross", (EXP,))) <= {BIN, ARITH} and set(code_sets.get("num_diff", (3,))) <= {1, 2}
            and n_picks >= 5)


def de_trial(key, pop, idx, best_index, pbest_rows, prim, sec, num_diff, F, CR, n_picks, lb, ub, cross=BIN, blocks=1,
             keys=None):
    """Trial vectors for the rows ``idx`` of ``pop``; ``key`` is split exactly as the PyTorch path does
    (selection / pbest / crossover → jrand, mask).  ``prim, sec, num_diff, cross`` are ints, or all four
    per-row integer tensors.

    ``blocks`` > 1: every row of ``pop`` is ``blocks`` independent sub-vectors (stacked members of a container);
    ``idx``, ``F``, ``CR`` and ``pbest_rows`` then run over the virtual rows ``individual * blocks + block``
    (``pbest_rows`` still names individuals), and the result has one row per entry of ``idx``; a ``best_index`` of
    ``blocks`` entries names the best individual of every block.  ``keys``: ``(selection, jrand, mask)`` keys already derived
    from ``key`` by the caller (static strategies), which saves the two split launches here."""
    dev = pop.device
    if best_index.numel() not in (1, blocks):
        raise ValueError("best_index: one individual, or one per block")
    if pop.stride(1) != 1 or (blocks > 1 and not pop.is_contiguous()):
        pop = pop.contiguous()
    n_pop = pop.shape[0]
    if blocks > 1:
        pop = pop.view(n_pop * blocks, pop.shape[1] // blocks)
    b, dim = idx.shape[0], pop.shape[1]
    per_row = isinstance(prim, torch.Tensor)
    if keys is not None and not per_row:
        k_sel, k_r, k_m = keys                               # derived by the caller (``random.split_nested(key, 3, 2, 2)``)
    else:
        k_sel, _k_pb, k_x = rnd.split(key, 3)
        if per_row:
            # the tensor formulation gives the binomial crossover the first half of a further split
            k_x = rnd.split(k_x)[0]
        k_r, k_m = rnd.split(k_x)
    if per_row:
        codes = torch.stack([prim, sec, num_diff, cross], dim=1).to(device=dev, dtype=torch.int32).contiguous()
        prim = sec = num_diff = cross = 0
    f, f_vec = _vec(F, b, dev)
    cr, cr_vec = _vec(CR, b, dev)
    out = torch.empty((b, dim), dtype=torch.float32, device=dev)
    check(hip().evox_hip_de_trial(
        pop.data_ptr(), pop.stride(0), n_pop, dim, idx.contiguous().data_ptr(), b,
        best_index.reshape(-1).contiguous().data_ptr(), pbest_rows.contiguous().data_ptr() if pbest_rows is not None else None,
        f.data_ptr(), f_vec, cr.data_ptr(), cr_vec, int(prim), int(sec), int(num_diff), int(cross),
        codes.data_ptr() if per_row else None, int(n_picks),
        k_sel.contiguous().data_ptr(), k_r.contiguous().data_ptr(), k_m.contiguous().data_ptr(),
        lb.contiguous().data_ptr(), ub.contiguous().data_ptr(), out.data_ptr(), int(blocks), int(best_index.numel() > 1), stream_of(pop)), "de_trial")
    return out
