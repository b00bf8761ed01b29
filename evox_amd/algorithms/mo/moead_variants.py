"""EAG-MOEA/D (Cai, Li & Fan 2015) and MOEA/D-M2M (Liu, Gu & Zhang 2014).
Reference: ``src/evox/algorithms/mo/eagmoead.py:42-212``, ``moeadm2m.py:21-203``."""
from __future__ import annotations

import math

import torch

from ...ops._util import count_groups, pick
from ...core import random as rnd
from ...core.state import State
from ...operators import crossover, selection
from ...operators.sampling import LatinHypercubeSampling, UniformSampling
from ...operators.selection.non_dominate import crowding_distance, non_dominated_sort
from ...ops import nds as _nds
from ...ops.distance import cdist as _cdist
from ...utils.common import AggregationFunction, cos_dist
from .common import MOAlgorithm, rank_crowding_truncate
from .moead import _scatter_best, two_distinct


class EAGMOEAD(MOAlgorithm):
    """MOEA/D working population guided by an external NSGA-II style archive: sub-problems whose
    offspring recently entered the archive (success history over the last ``LGs`` generations)
    are given more offspring.  The neighbourhood replacement of the working population is the
    scatter-min form described in ``moead.py`` (an offspring replaces every neighbour it is at
    least as good as; the best offspring wins a contested sub-problem).

    ``aggregation``: the scalarising function of the working population.  The reference hard-codes ``"weighted_sum"``
    (``eagmoead.py:79``; the algorithm was proposed for combinatorial problems), under which the working population of a
    problem with a concave front collapses onto the corners of the front and the archive is fed from there only.  The
    default here is ``"tchebycheff"`` against the running ideal point, which reaches the interior of such fronts;
    ``aggregation="weighted_sum"`` reproduces the reference."""

    def __init__(self, lb, ub, n_objs, pop_size, LGs=8, selection_op=None, mutation_op=None, crossover_op=None,
                 aggregation: str = "tchebycheff"):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op,
                      crossover_op if crossover_op is not None else crossover.SimulatedBinary(type=2),
                      selection_op if selection_op is not None else selection.RouletteWheelSelection(int(pop_size)))
        self.LGs = LGs
        self.T = math.ceil(self.pop_size / 10)
        self.sample = LatinHypercubeSampling(self.pop_size, self.n_objs)
        if aggregation not in ("weighted_sum", "tchebycheff"):
            raise ValueError("aggregation must be 'weighted_sum' or 'tchebycheff'")
        self.aggregation = aggregation
        self.aggregate_func = AggregationFunction(aggregation)

    def setup(self, key):
        f = self._base_fields(key)
        key, k_w = rnd.split(f["key"])
        f["key"] = key
        w = self.sample(k_w)
        B = torch.argsort(_cdist(w, w), dim=1, stable=True)[:, : self.T]
        n, dev = self.pop_size, self.device
        return State(inner_pop=f["population"], inner_obj=f["fitness"], weight_vector=w, B=B,
                     s=torch.zeros((n, self.LGs), device=dev), offspring_loc=torch.zeros(n, dtype=torch.int64, device=dev),
                     ideal=torch.full((self.n_objs,), float("inf"), device=dev), gen=0, **f)

    def init_tell(self, state, fitness):
        return state.replace(fitness=fitness, inner_obj=fitness, ideal=fitness.amin(0))

    def ask(self, state):
        key, k_perm, k_sel, k_x, k_mut = rnd.split(state.key, 5)
        n, t = state.B.shape
        s = state.s.sum(1) + 1e-6
        d = s / s.sum() + 0.002
        d = d / d.sum()
        _, loc = self.selection(k_sel, state.inner_pop, 1.0 / d)
        first, second = two_distinct(k_perm, n, t)               # the first two of a random permutation of each row of B
        parent = torch.gather(state.B, 1, torch.stack([first, second], dim=1))[loc]
        mates = torch.cat([state.inner_pop[parent[:, 0]], state.inner_pop[parent[:, 1]]], dim=0)
        nxt = self._vary(k_x, k_mut, mates)
        return nxt, state.replace(next_generation=nxt, offspring_loc=loc, key=key)

    # a recorded generation: ``gen`` only selects the column of the success history it rewrites: every column has its
    # own recording (``graph_variant``), and the counter itself stays the generation count, advanced on the host
    def graph_host_counters(self):
        return ("gen",)

    def graph_variant(self, state):
        return (state.gen + 1) % self.LGs

    def tell(self, state, fitness):
        gen = state.gen + 1
        n = self.pop_size
        w, B, loc = state.weight_vector, state.B, state.offspring_loc
        nb = B[loc]                                                                 # (n, T) targets of offspring i
        ideal = torch.minimum(state.ideal, fitness.amin(0))
        if self.aggregation == "weighted_sum":
            g_old = self.aggregate_func(state.inner_obj, w)
            g_new = (fitness[:, None, :] * w[nb]).sum(-1)
        else:
            g_old = self.aggregate_func(state.inner_obj, w, ideal)
            g_new = ((fitness[:, None, :] - ideal).abs() * w[nb]).amax(-1)
        bar = torch.nextafter(g_old, torch.full_like(g_old, float("inf")))          # ">=" replaces
        _, winner = _scatter_best(g_new, nb, n, bar)
        take = winner >= 0
        src = winner.clamp_min(0)
        inner_pop = torch.where(take[:, None], state.next_generation[src], state.inner_pop)
        inner_obj = torch.where(take[:, None], fitness[src], state.inner_obj)
        pop, obj = self._merged(state, fitness)
        keep = rank_crowding_truncate(pop, obj, n)
        entered = keep >= n
        hist = torch.zeros(n, device=obj.device).scatter_add(0, loc[(keep - n).clamp_min(0)], entered.to(torch.float32))
        # the success history changes only in generations in which an offspring entered the archive (no read-back:
        # the column is rewritten with its old content otherwise)
        col = gen % self.LGs
        s = state.s.clone()
        s[:, col] = torch.where(entered.any(), hist, state.s[:, col])
        return state.replace(population=pop[keep], fitness=obj[keep], inner_pop=inner_pop, inner_obj=inner_obj, s=s,
                             ideal=ideal, gen=gen)


class M2MCrossover:
    def __call__(self, key, p1, p2, scale):
        k1, k2 = rnd.split(key)
        n = p1.shape[0]
        rc = (2 * rnd.uniform(k1, (n, 1)) - 1) * (1 - rnd.uniform(k2, (n, 1))) ** (-((1 - scale) ** 0.7))
        return p1 + rc * (p1 - p2)


class M2MMutation:
    def __call__(self, key, p1, off, scale, lb, ub):
        k1, k2, k3, k4 = rnd.split(key, 4)
        n, d = p1.shape
        rm = 0.25 * (2 * rnd.uniform(k1, (n, d)) - 1) * (1 - rnd.uniform(k2, (n, d))) ** (-((1 - scale) ** 0.7))
        site = rnd.uniform(k3, (n, d)) < 1 / d
        out = torch.where(site, off + rm * (ub - lb), off)
        r = rnd.uniform(k4, (n, d))
        out = torch.where(out < lb, lb + 0.5 * r * (p1 - lb), out)
        out = torch.where(out > ub, ub - 0.5 * r * (ub - p1), out)
        return out


def m2m_associate(key, pop, obj, w, s):
    """Split the population into ``len(w)`` sub-populations of ``s`` members by the closest
    direction; under-full regions are topped up with random individuals, over-full ones keep
    their best ``s`` by (front index, crowding distance).  All regions are handled together: one
    lexicographic sort by (region, front index, −crowding distance) and a gather of the first ``s``
    positions of every region, without reading a count back to the host."""
    k, n = w.shape[0], pop.shape[0]
    dev = pop.device
    region = torch.argmax(cos_dist(obj, w), dim=1)
    rank = non_dominated_sort(obj)
    filler = rnd.randint(key, (k, s), 0, n)
    cnt = count_groups(region, k)
    start = torch.cumsum(cnt, 0) - cnt
    by_rank = torch.argsort(rank, stable=True)
    by_rank = by_rank[torch.argsort(region[by_rank], stable=True)]          # grouped by region, ascending front index
    # front index of each region's s-th member: the front whose members compete by crowding distance
    worst = rank[by_rank[(start + s - 1).clamp_max(n - 1)]]
    in_worst = (rank == worst[region]) & (cnt[region] >= s)
    cd = _nds.segmented_crowding_distance(obj, region, in_worst, k)
    tie = torch.where(in_worst, -cd, torch.zeros_like(cd))
    order = torch.argsort(tie, stable=True)
    order = order[torch.argsort(rank[order], stable=True)]
    order = order[torch.argsort(region[order], stable=True)]                # (region, front index, −crowding)
    j = torch.arange(s, device=dev)[None, :]
    have = j < cnt[:, None]
    own = order[(start[:, None] + j).clamp_max(n - 1)]
    fill = torch.gather(filler, 1, (j - cnt[:, None]).clamp_min(0))
    part = torch.where(have, own, fill).reshape(-1)
    return pop[part], obj[part]


def m2m_associate_reference(key, pop, obj, w, s):
    """``m2m_associate`` region by region (host loop; the oracle of the batched form)."""
    k, n = w.shape[0], pop.shape[0]
    region = torch.argmax(cos_dist(obj, w), dim=1)
    rank = non_dominated_sort(obj).to(torch.float32)
    filler = rnd.randint(key, (k, s), 0, n)
    chosen = []
    for i in range(k):
        mask = region == i
        cnt = int(mask.sum())
        if cnt < s:
            members = torch.nonzero(mask)[:, 0]
            chosen.append(torch.cat([members, filler[i, : s - cnt]]))
        else:
            r = torch.where(mask, rank, torch.full_like(rank, float("inf")))
            order = torch.argsort(r, stable=True)
            worst = pick(r, order[s - 1])
            cd = crowding_distance(obj, r == worst)
            tie = torch.where(r == worst, -cd, torch.zeros_like(cd))
            idx = torch.argsort(tie, stable=True)
            chosen.append(idx[torch.argsort(r[idx], stable=True)][:s])
    part = torch.cat(chosen)
    return pop[part], obj[part]


class MOEADM2M(MOAlgorithm):
    def __init__(self, lb, ub, n_objs, pop_size, k=10, max_gen=100, mutation_op=None, crossover_op=None, seed=0):
        self._init_mo(lb, ub, n_objs, pop_size)
        self.max_gen = max_gen
        self.mutation = M2MMutation()
        self.crossover = M2MCrossover()
        w, kk = UniformSampling(k, self.n_objs)()
        self.w = w.to(self.device)
        self.k = int(kk)
        self.s = int(self.pop_size / self.k)
        self.pop_size = self.s * self.k

    def setup(self, key):
        return State(gen=0, **self._base_fields(key))

    def ask(self, state):
        key, k_loc, k_glob, k_r, k_x, k_mut = rnd.split(state.key, 6)
        # past ``max_gen`` the operators keep their final step distribution (the reference's (1 − scale)^0.7 turns
        # NaN there, ``moeadm2m.py:29,42``)
        scale = min(state.gen / self.max_gen, 1.0)
        s, k, n = self.s, self.k, self.pop_size
        base = torch.arange(0, s * k, s, device=self.device).repeat_interleave(s)
        local = rnd.randint(k_loc, (n,), 0, s) + base
        anywhere = rnd.randint(k_glob, (n,), 0, n)
        mate = torch.where(rnd.uniform(k_r, (n,)) < 0.7, anywhere, local)
        pop = state.population
        child = self.crossover(k_x, pop, pop[mate], scale)
        nxt = self.mutation(k_mut, pop, child, scale, self.lb, self.ub)
        return nxt, state.replace(next_generation=nxt, key=key, gen=state.gen + 1)

    def init_tell(self, state, fitness):
        key, sub = rnd.split(state.key)
        pop, obj = m2m_associate(sub, state.population, fitness, self.w, self.s)
        return state.replace(population=pop, fitness=obj, key=key)

    def tell(self, state, fitness):
        key, sub = rnd.split(state.key)
        pop, obj = self._merged(state, fitness)
        pop, obj = m2m_associate(sub, pop, obj, self.w, self.s)
        return state.replace(population=pop, fitness=obj, key=key)
