"""NSGA-III (Deb & Jain 2014).  Reference: ``src/evox/algorithms/mo/nsga3.py:27-211``.

The reference fills the last front with a loop that admits one individual per iteration.  The
same selection is computed here without a population-length sequential loop: the loop always
serves the reference direction with the fewest members, so the final niche counts are a
*water-filling* of the remaining slots — the smallest level L with
Σ_j min(c_j, max(0, L − ρ_j)) ≥ K, found by bisection — and inside a niche the loop takes the
closest candidate first when the niche was empty and random candidates otherwise, i.e. a prefix
of a (partly) random order.  Ties between niches at the last level go to the lower index, as
``argmin`` does."""
from __future__ import annotations

import torch

from ...ops import moselect as _sel
from ...ops._util import count_groups, pick
from ... import _native
from ...core import random as rnd
from ...core.state import State
from ...operators.sampling import UniformSampling
from ...operators.selection.non_dominate import non_dominated_sort
from ...utils.common import cos_dist
from ...ops._util import check as _check
from ...ops._util import hip as _hip
from ...ops._util import stream_of as _stream_of
from ...ops.linalg_small import solve_small
from .common import MOAlgorithm

_native.register_hip_api({
    "evox_hip_niche_fill": (_native.I, [_native.P, _native.P, _native.I, _native.P, _native.P, _native.P]),
})


def _hyperplane_intercepts(extreme):
    """``(intercepts, ok)`` of the hyperplane through the rows of ``extreme`` (m × m): the solution of
    ``extreme · x = 1`` inverted, and whether the system was well posed."""
    sol, ok = solve_small(extreme, torch.ones(extreme.shape[0], dtype=extreme.dtype, device=extreme.device))
    return 1.0 / sol, ok


def _niche_fill(rho, cnt, k, max_level: int = 1 << 40):
    """Slots t_j granted to each niche (ρ_j members already, c_j candidates) when ``k`` more
    individuals are admitted one at a time to the currently least crowded niche.  ``max_level``:
    a host-known bound on ρ_j + c_j + 1 (the population size), which bounds the bisection."""
    rho = rho.to(torch.int64)
    cnt = cnt.to(torch.int64)
    if rho.is_cuda:
        # the whole bisection and the tie-breaking prefix count in one single-workgroup launch
        kk = torch.as_tensor(k, dtype=torch.int64, device=rho.device).reshape(1)
        rho, cnt = rho.contiguous(), cnt.contiguous()
        out = torch.empty_like(rho)
        _check(_hip().evox_hip_niche_fill(rho.data_ptr(), cnt.data_ptr(), rho.shape[0], kk.data_ptr(), out.data_ptr(),
                                          _stream_of(rho)), "niche_fill")
        return out
    lo = torch.zeros((), dtype=torch.int64, device=rho.device)
    hi = (rho.max() + cnt.max() + 1).to(torch.int64)

    def granted(level):
        return torch.minimum(cnt, (level - rho).clamp_min(0))

    for _ in range(max_level.bit_length() + 1):        # bisection on the level (an integer below max_level)
        mid = (lo + hi) // 2
        enough = granted(mid).sum() >= k
        hi = torch.where(enough, mid, hi)
        lo = torch.where(enough, lo, mid + 1)
    level = hi
    base = granted(level - 1)
    extra_ok = granted(level) > base
    need = k - base.sum()
    order = torch.cumsum(extra_ok.to(torch.int64), 0)
    return base + (extra_ok & (order <= need)).to(torch.int64)


class NSGA3(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, selection_op=None, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op, selection_op)
        self.ref = UniformSampling(self.pop_size, self.n_objs)()[0].to(self.device)

    def ask(self, state):
        key, mut_key, x_key = rnd.split(state.key, 3)
        nxt = self._vary(x_key, mut_key, state.population)
        return nxt, state.replace(next_generation=nxt, key=key)

    def _normalise(self, fit, considered):
        m = self.n_objs
        inf = torch.full_like(fit, float("inf"))
        ideal = torch.where(considered[:, None], fit, inf).amin(0)
        shifted = fit - ideal
        # extreme point of axis i: minimiser of the achievement scalarising function
        w = torch.eye(m, device=fit.device) + 1e-6
        asf = (shifted[None, :, :] / w[:, None, :]).amax(-1)                       # (m, n)
        asf = torch.where(considered[None, :], asf, torch.full_like(asf, float("inf")))
        extreme = shifted[torch.argmin(asf, dim=1)]                                # (m, m)
        worst = torch.where(considered[:, None], shifted, -inf).amax(0)
        # hyperplane through the extreme points; fall back to the worst point when degenerate
        intercept, full_rank = _hyperplane_intercepts(extreme)
        ok = full_rank & torch.isfinite(intercept).all() & (intercept > 1e-12).all()
        nadir = torch.where(ok, intercept, worst).clamp_min(1e-12)
        return shifted / nadir

    def tell(self, state, fitness):
        key, k_sel = rnd.split(state.key)
        pop, fit = self._merged(state, fitness)
        n, k_total = fit.shape[0], self.pop_size
        rank = non_dominated_sort(fit, until=k_total)
        order = torch.argsort(rank, stable=True)
        last_rank = pick(rank, order[k_total - 1])
        considered = rank <= last_rank
        norm = self._normalise(fit, considered)
        # niche of every point: the reference line at the smallest perpendicular distance (streamed on the GPU)
        group, _, gdist = _sel.ref_associate(norm, self.ref)
        nref = self.ref.shape[0]
        earlier, last = rank < last_rank, rank == last_rank
        rho = torch.zeros(nref, dtype=torch.int64, device=fit.device).scatter_add(0, group, earlier.to(torch.int64))
        cnt = torch.zeros(nref, dtype=torch.int64, device=fit.device).scatter_add(0, group, last.to(torch.int64))
        need = k_total - earlier.sum()
        slots = _niche_fill(rho, cnt, need, max_level=2 * n + 2)                   # (nref,)
        # order of the candidates inside their niche: closest first if the niche was empty,
        # otherwise random
        u = rnd.uniform(k_sel, (n,))
        closest = torch.full((nref,), float("inf"), device=fit.device).scatter_reduce(
            0, group, torch.where(last, gdist, torch.full_like(gdist, float("inf"))), reduce="amin")
        is_closest = last & (gdist == closest[group]) & (rho[group] == 0)
        inner = torch.where(is_closest, torch.full_like(u, -1.0), u)
        inner = torch.where(last, inner, torch.full_like(u, 2.0))
        by_inner = torch.argsort(inner, stable=True)
        by_group = by_inner[torch.argsort(group[by_inner], stable=True)]           # grouped, inner-sorted
        g_sorted = group[by_group]
        start = torch.zeros(nref + 1, dtype=torch.int64, device=fit.device)
        start[1:] = torch.cumsum(count_groups(g_sorted, nref), 0)
        pos_sorted = torch.arange(n, device=fit.device) - start[g_sorted]
        pos = torch.empty(n, dtype=torch.int64, device=fit.device)
        pos[by_group] = pos_sorted
        chosen = earlier | (last & (pos < slots[group]))
        idx = torch.argsort((~chosen).to(torch.int8), stable=True)[:k_total]
        return state.replace(population=pop[idx], fitness=fit[idx], key=key)
