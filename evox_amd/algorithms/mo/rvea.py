"""RVEA (Cheng et al. 2016) and RVEAa (RVEA with an additional, regenerated reference-vector
set).  Reference: ``src/evox/algorithms/mo/rvea.py:16-140``, ``rveaa.py:18-215``.  Populations
keep a fixed size equal to the number of reference vectors; vectors without an associated
solution hold NaN rows."""
from __future__ import annotations

import torch

from ...core import random as rnd
from ...core.state import State
from ...operators.sampling import UniformSampling
from ...operators.selection.non_dominate import non_dominated_sort
from ...operators.selection.rvea_selection import ReferenceVectorGuided, ref_vec_guided
from ...ops import moselect as _sel
from ...utils.common import cos_dist
from .common import MOAlgorithm


def _nan_amax(x):
    return torch.where(torch.isnan(x), torch.full_like(x, float("-inf")), x).amax(0)


def _nan_amin(x):
    return torch.where(torch.isnan(x), torch.full_like(x, float("inf")), x).amin(0)


def _valid_first(population):
    """Indices that list the non-NaN rows first (stable), and how many there are."""
    ok = ~torch.isnan(population).all(dim=1)
    order = torch.argsort((~ok).to(torch.int8), stable=True)
    return order, ok.sum()


class RVEA(MOAlgorithm):
    """``alpha`` controls how fast the angle penalty grows, ``fr`` is the frequency of
    reference-vector adaptation, ``max_gen`` the planned number of generations."""

    def __init__(self, lb, ub, n_objs, pop_size, alpha=2, fr=0.1, max_gen=100, selection_op=None, mutation_op=None,
                 crossover_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op, crossover_op,
                      selection_op if selection_op is not None else ReferenceVectorGuided())
        self.alpha, self.fr, self.max_gen = alpha, fr, max_gen
        self.sampling = UniformSampling(self.pop_size, self.n_objs)
        v, n = self.sampling()
        self._v0 = v.to(self.device)
        self.pop_size = n

    def setup(self, key):
        f = self._base_fields(key)
        # the generation number twice: as a host integer (which generations adapt the vectors is control flow) and on the
        # device, where the angle-penalty weight is computed from it — a recorded generation then follows the counter
        return State(reference_vector=self._v0, init_v=self._v0, gen=0, gen_dev=torch.zeros((), device=self.device), **f)

    def _theta(self, gen_dev):
        return (gen_dev / self.max_gen) ** self.alpha

    # recorded generations: the host counter only decides which generations adapt the reference vectors
    def graph_host_counters(self):
        return ("gen",)                 # stays the generation count; the adapting generations are a variant of their own

    def graph_variant(self, state):
        period = round(1 / self.fr)
        return period > 0 and (state.gen + 1) % period == 0

    def _mating_pool(self, key, population, n):
        order, count = _valid_first(population)
        pick = (rnd.uniform(key, (n,)) * count).to(torch.int64).clamp_(max=population.shape[0] - 1)
        return population[order[pick]]

    def ask(self, state):
        key, k_pool, x_key, mut_key = rnd.split(state.key, 4)
        parents = self._mating_pool(k_pool, state.population, self.pop_size)
        nxt = self._vary(x_key, mut_key, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def _adapt(self, gen, fitness, v, v0):
        period = round(1 / self.fr)
        if period > 0 and gen % period == 0:
            return v0 * (_nan_amax(fitness) - _nan_amin(fitness))
        return v

    def tell(self, state, fitness):
        gen, gen_dev = state.gen + 1, state.gen_dev + 1
        pop, fit = self._merged(state, fitness)
        survivors, survivor_fit = self.selection(pop, fit, state.reference_vector, self._theta(gen_dev))
        v = self._adapt(gen, survivor_fit, state.reference_vector, state.init_v)
        return state.replace(population=survivors, fitness=survivor_fit, reference_vector=v, gen=gen, gen_dev=gen_dev)


def rv_regeneration(pop_obj, v, key):
    """Re-draw the additional vectors that lost all their solutions: uniform in the box spanned
    by the current objective ranges."""
    obj = pop_obj - _nan_amin(pop_obj)
    if obj.is_cuda and 2 <= obj.shape[1] <= 8 and obj.dtype == torch.float32 and v.dtype == torch.float32:
        # nearest vector by angle, streamed (objectives and vectors are non-negative here, so the smallest perpendicular
        # distance is the largest cosine); rows without a solution go to vector 0, as below
        associate = _sel.ref_associate(obj, v)[0]
    else:
        cosine = cos_dist(torch.where(torch.isnan(obj), torch.zeros_like(obj), obj), v)
        cosine = torch.where(torch.isnan(pop_obj).any(1, keepdim=True), torch.full_like(cosine, -1.0), cosine)
        associate = torch.argmax(cosine, dim=1)
    used = torch.zeros(v.shape[0], dtype=torch.bool, device=v.device).index_fill_(0, associate, True)
    fresh = rnd.uniform(key, v.shape) * _nan_amax(obj)
    return torch.where(used[:, None], v, fresh)


def batch_truncation(pop, obj, k):
    """Remove the ``n − k`` individuals with the largest nearest-neighbour cosine similarity
    (computed once for the batch)."""
    n = obj.shape[0]
    if n <= k:
        return pop, obj
    c = cos_dist(obj, obj)
    c = c.masked_fill(torch.eye(n, dtype=torch.bool, device=obj.device), -2.0)
    c = torch.where(torch.isnan(c), torch.full_like(c, -2.0), c)
    crowded = c.amax(1)
    crowded = torch.where(torch.isnan(obj).any(1), torch.full_like(crowded, 2.0), crowded)
    keep = torch.argsort(crowded, stable=True)[:k]
    return pop[keep], obj[keep]


class RVEAa(RVEA):
    """RVEA* of the RVEA paper: a second set of randomly regenerated reference vectors covers
    irregular fronts; the final generation keeps only first-front members and truncates by
    angular crowding."""

    def setup(self, key):
        f = self._base_fields(key)
        key, k_v = rnd.split(f["key"])
        f["key"] = key
        extra = rnd.uniform(k_v, self._v0.shape)
        v = torch.cat([self._v0, extra], dim=0)
        n2 = v.shape[0]
        pop = torch.cat([f["population"], torch.full_like(f["population"], float("nan"))], dim=0)
        f["population"] = pop
        f["fitness"] = torch.zeros((n2, self.n_objs), device=self.device)
        f["next_generation"] = f["next_generation"]
        return State(reference_vector=v, init_v=self._v0, gen=0, gen_dev=torch.zeros((), device=self.device), **f)

    def init_ask(self, state):
        return state.population[: self.pop_size], state

    def init_tell(self, state, fitness):
        fit = torch.cat([fitness, torch.full_like(fitness, float("nan"))], dim=0)
        return state.replace(fitness=fit)

    def ask(self, state):
        key, k_pool, x_key, mut_key = rnd.split(state.key, 4)
        parents = self._mating_pool(k_pool, state.population, self.pop_size)
        nxt = self._vary(x_key, mut_key, parents)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        gen, gen_dev = state.gen + 1, state.gen_dev + 1
        key, k_rv = rnd.split(state.key)
        pop = torch.cat([state.population, state.next_generation], dim=0)
        fit = torch.cat([state.fitness, fitness], dim=0)
        # only first-front members compete for the vectors
        ranked = torch.where(torch.isnan(fit).any(1, keepdim=True), torch.full_like(fit, float("inf")), fit)
        front = non_dominated_sort(ranked, until=1) == 0
        fit_f = torch.where(front[:, None], fit, torch.full_like(fit, float("nan")))
        pop_f = torch.where(front[:, None], pop, torch.full_like(pop, float("nan")))
        v = state.reference_vector
        survivors, survivor_fit = ref_vec_guided(pop_f, fit_f, v, self._theta(gen_dev))
        n0 = self._v0.shape[0]
        v_main = self._adapt(gen, survivor_fit, v[:n0], state.init_v)
        v_extra = rv_regeneration(survivor_fit, v[n0:], k_rv)
        return state.replace(population=survivors, fitness=survivor_fit, reference_vector=torch.cat([v_main, v_extra]),
                             gen=gen, gen_dev=gen_dev, key=key)

    def final_population(self, state):
        """First-front, angularly truncated population of size ``pop_size`` (call after the run)."""
        ok = ~torch.isnan(state.fitness).any(1)
        return batch_truncation(state.population[ok], state.fitness[ok], self.pop_size)
