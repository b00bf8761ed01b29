"""KnEA (Zhang, Tian & Jin 2015), BiGE (Li, Yang & Liu 2015) and θ-DEA (Yuan et al. 2016).
Reference: ``src/evox/algorithms/mo/knea.py:38-221``, ``bige.py:26-142``, ``tdea.py:22-178``."""
from __future__ import annotations

import torch

from ...ops._util import count_groups, pick
from ...core import random as rnd
from ...core.state import State
from ...operators import selection
from ...operators.sampling import UniformSampling
from ...operators.selection.non_dominate import non_dominated_sort
from ...ops import moselect as _sel
from ...ops.linalg_small import solve_small
from ...utils.common import cos_dist
from .common import MOAlgorithm


def calc_DW(fit, k):
    """Weighted distance to the k nearest neighbours (KnEA's density estimate)."""
    nb = _sel.knn_distances(fit, k)
    avg = nb.sum(1) / k
    r = 1 / (nb - avg[:, None]).abs().clamp_min(1e-12)
    w = r / r.sum(1, keepdim=True)
    return (nb * w).sum(1)


class KnEA(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, knee_rate=0.5, k_neighbors=3, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.knee_rate, self.k_neighbors = knee_rate, k_neighbors
        self.selection = selection.Tournament(self.pop_size, multi_objective=True)

    def setup(self, key):
        dev = self.device
        return State(knee=torch.zeros(self.pop_size, dtype=torch.bool, device=dev), r=torch.tensor(1.0, device=dev),
                     t=torch.tensor(0.0, device=dev), **self._base_fields(key))

    def ask(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        rank = non_dominated_sort(state.fitness).to(torch.float32)
        DW = calc_DW(state.fitness, self.k_neighbors)
        # lexicographic tournament: front index first, then knee membership, then larger DW
        parents, _ = self.selection(k_sel, state.population, -DW, (~state.knee).to(torch.float32), rank)
        nxt = self._vary(k_x, k_mut, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def _plane(self, f):
        m = self.n_objs
        extreme = f[torch.argmax(f, dim=0)]
        sol, ok = solve_small(extreme, torch.ones(m, device=f.device))
        return torch.where(ok, sol, 1.0 / torch.diagonal(extreme).clamp_min(1e-6))

    def tell(self, state, fitness):
        pop, fit = self._merged(state, fitness)
        n = fit.shape[0]
        rank = non_dominated_sort(fit, until=self.pop_size)
        order = torch.argsort(rank, stable=True)
        rank, pop, fit = rank[order], pop[order], fit[order]
        last_rank = int(rank[self.pop_size - 1])
        knee = torch.zeros(n, dtype=torch.bool, device=fit.device)
        r, t = state.r, state.t
        plane = torch.ones(self.n_objs, device=fit.device)
        for i in range(last_rank + 1):
            members = torch.nonzero(rank == i)[:, 0]
            f = fit[members]
            plane = self._plane(f)
            far_first = torch.argsort(plane @ f.T, stable=True)              # closest to the ideal side first
            r = r * torch.exp(-(1 - t / self.knee_rate) / self.n_objs)
            R = (f.amax(0) - f.amin(0)) * r
            k_i = _sel.knee_suppress(f, far_first, R)
            knee[members] = k_i
            t = k_i.sum() / f.shape[0]
        selected = (rank < last_rank) | knee
        dif = int(selected.sum()) - self.pop_size
        in_last = rank == last_rank
        score = fit @ plane
        if dif > 0:          # too many: drop the knees of the last front farthest along the plane normal
            cand = knee & in_last
            worst = torch.argsort(torch.where(cand, -score, torch.full_like(score, float("inf"))), stable=True)[:dif]
            selected[worst] = False
        elif dif < 0:        # too few: add the non-knee members of the last front closest to the plane
            cand = ~knee & in_last
            best = torch.argsort(torch.where(cand, score, torch.full_like(score, float("inf"))), stable=True)[:-dif]
            selected[best] = True
        idx = torch.argsort((~selected).to(torch.int8), stable=True)[: self.pop_size]
        return state.replace(population=pop[idx], fitness=fit[idx], knee=knee[idx], r=r, t=t)


def bige_estimate(fit, mask):
    """Bi-goal values of the rows selected by ``mask``: proximity (sum of normalised
    objectives) and crowding degree (sharing function over a niche of radius n^(-1/m))."""
    m = fit.shape[1]
    n = mask.sum().clamp_min(1)
    r = 1 / n.to(torch.float32) ** (1 / m)
    big = torch.full_like(fit, float("inf"))
    f_max = torch.where(mask[:, None], fit, -big).amax(0)
    f_min = torch.where(mask[:, None], fit, big).amin(0)
    normed = (fit - f_min) / (f_max - f_min).clamp_min(1e-6)
    normed = torch.where(mask[:, None], normed, torch.full_like(normed, float(m)))
    pr = normed.sum(1)
    cd = _sel.bige_crowding(normed, pr, r)             # streaming kernel on the GPU: no n × n matrices
    bi = torch.stack([pr, cd], dim=1)
    return torch.where(mask[:, None], bi, torch.full_like(bi, float("inf")))


class BiGE(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self.selection = selection.Tournament(self.pop_size)

    def ask(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        everyone = torch.ones(self.pop_size, dtype=torch.bool, device=self.device)
        # (the reference feeds the decision vectors to the estimator here, ``bige.py:105``;
        # the bi-goal transformation is defined on objective values)
        bi_rank = non_dominated_sort(bige_estimate(state.fitness, everyone))
        parents, _ = self.selection(k_sel, state.population, bi_rank.to(torch.float32))
        nxt = self._vary(k_x, k_mut, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, fit = self._merged(state, fitness)
        rank = non_dominated_sort(fit, until=self.pop_size)
        order = torch.argsort(rank, stable=True)
        rank, pop, fit = rank[order], pop[order], fit[order]
        last_rank = rank[self.pop_size - 1]
        in_last = rank == last_rank
        bi_rank = non_dominated_sort(bige_estimate(fit, in_last))
        final = torch.where(rank < last_rank, torch.full_like(bi_rank, -1),
                            torch.where(in_last, bi_rank, torch.full_like(bi_rank, fit.shape[0] + 1)))
        idx = torch.argsort(final, stable=True)[: self.pop_size]
        return state.replace(population=pop[idx], fitness=fit[idx])


def theta_nd_sort(obj, w, mask):
    """θ-non-dominated level of every (masked) point: points are clustered to their nearest
    reference direction and ranked inside the cluster by d₁ + θ d₂."""
    n, nw = obj.shape[0], w.shape[0]
    cls, d1, d2 = _sel.ref_associate(obj, w)          # nearest direction by perpendicular distance (streamed on the GPU)
    theta = torch.full((nw,), 5.0, device=obj.device)
    theta = torch.where((w > 1e-4).sum(1) == 1, torch.full_like(theta, 1e6), theta)
    ar = torch.arange(n, device=obj.device)
    score = d1 + theta[cls] * d2
    score = torch.where(mask, score, torch.full_like(score, float("inf")))
    # rank inside each cluster = position among the cluster's members sorted by score
    by_score = torch.argsort(score, stable=True)
    grouped = by_score[torch.argsort(cls[by_score], stable=True)]
    g = cls[grouped]
    start = torch.zeros(nw + 1, dtype=torch.int64, device=obj.device)
    start[1:] = torch.cumsum(count_groups(g, nw), 0)
    level = torch.empty(n, dtype=torch.int64, device=obj.device)
    level[grouped] = ar - start[g] + 1
    return torch.where(mask, level.to(torch.float32), torch.full((n,), float("inf"), device=obj.device))


class TDEA(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, mutation_op=None, crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op)
        self._w = UniformSampling(self.pop_size, self.n_objs)()[0].to(self.device)

    def setup(self, key):
        z = torch.zeros(self.n_objs, device=self.device)
        return State(w=self._w, z=z, z_nad=z.clone(), **self._base_fields(key))

    def init_tell(self, state, fitness):
        return state.replace(fitness=fitness, z=fitness.amin(0), z_nad=fitness.amax(0))

    def ask(self, state):
        key, k_sel, k_x, k_mut = rnd.split(state.key, 4)
        pool = rnd.randint(k_sel, (self.pop_size,), 0, self.pop_size)
        nxt = self._vary(k_x, k_mut, state.population[pool])
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, obj = self._merged(state, fitness)
        m, n = self.n_objs, self.pop_size
        rank = non_dominated_sort(obj, until=n)
        order = torch.argsort(rank, stable=True)
        mask = rank <= pick(rank, order[n - 1])
        z = torch.minimum(state.z, obj.amin(0))
        w1 = torch.full((m, m), 1e-6, device=obj.device) + torch.eye(m, device=obj.device) * (1 - 1e-6)
        scaled = ((obj - z) / (state.z_nad - z)).abs()
        asf = (scaled[:, None, :] / w1[None, :, :]).amax(-1)                       # (N, m)
        extreme = torch.argmin(asf, dim=0)
        A = obj[extreme] - z
        sol, _ = solve_small(A, torch.ones(m, device=obj.device))
        a = z + 1 / sol
        bad = torch.isnan(a).any() | (a <= z).any() | ~torch.isfinite(a).all()
        z_nad = torch.where(bad, obj.amax(0), a)
        norm_obj = (obj - z) / (z_nad - z)
        t_rank = theta_nd_sort(norm_obj, state.w, mask)
        idx = torch.argsort(t_rank, stable=True)
        idx = idx[torch.argsort(rank[idx], stable=True)][:n]
        return state.replace(population=pop[idx], fitness=obj[idx], z=z, z_nad=z_nad)
