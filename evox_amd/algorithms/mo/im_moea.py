"""IM-MOEA: inverse modelling with Gaussian processes (Cheng, Jin, Narukawa & Sendhoff 2015).
Reference: ``src/evox/algorithms/mo/im_moea.py:55-353``.

The population is partitioned by ``k`` reference directions.  In every sub-population, for each
objective m a random share of the members trains one-dimensional *inverse* models
f_m → x_d for ``l`` random decision variables; offspring are sampled from the models'
predictive distribution on an even grid of objective values, then mutated.

The reference fits K·M·l separate GPs with a linear kernel by 250 Adam steps each.  A zero-mean
GP with kernel σ²(x·x' + 1) on a scalar input is Bayesian linear regression on the features
(x, 1); with a vague prior its evidence-maximising noise level and its predictive distribution
have closed forms in the 2×2 normal equations.  All K·M·l models are therefore solved in one
batch from masked sufficient statistics – no iterative fitting and no per-model launches."""
from __future__ import annotations

import math

import torch

from ...core import random as rnd
from ...core.state import State
from ...operators.sampling import UniformSampling
from ...utils.common import cos_dist
from .common import MOAlgorithm, rank_crowding_truncate


def inverse_models(x, y, mask, x_new, ridge=1e-6):
    """Batched 1-D Bayesian linear regression.  ``x`` (G,P) inputs, ``y`` (G,P,L) targets,
    ``mask`` (G,P) valid rows, ``x_new`` (G,P) query points → predictive (mean, std), both
    (G,P,L)."""
    w = mask.to(x.dtype)
    n = w.sum(1).clamp_min(3.0)
    sx, sxx = (w * x).sum(1), (w * x * x).sum(1)
    sy = (w[:, :, None] * y).sum(1)
    sxy = ((w * x)[:, :, None] * y).sum(1)
    syy = (w[:, :, None] * y * y).sum(1)
    lam = ridge * (sxx + n)
    a11, a12, a22 = sxx + lam, sx, n + lam
    det = (a11 * a22 - a12 * a12).clamp_min(1e-30)
    i11, i12, i22 = a22 / det, -a12 / det, a11 / det
    slope = i11[:, None] * sxy + i12[:, None] * sy
    bias = i12[:, None] * sxy + i22[:, None] * sy
    rss = (syy - slope * sxy - bias * sy).clamp_min(0)
    noise = rss / (n - 2)[:, None]
    mean = slope[:, None, :] * x_new[:, :, None] + bias[:, None, :]
    lever = i11[:, None] * x_new * x_new + 2 * i12[:, None] * x_new + i22[:, None]
    std = torch.sqrt(noise[:, None, :] * (1 + lever.clamp_min(0))[:, :, None])
    return mean, std


class IMMOEA(MOAlgorithm):
    def __init__(self, lb, ub, n_objs=3, pop_size=105, l=3, k=10, mutation_op=None):
        self._init_mo(lb, ub, n_objs, pop_size, mutation_op)
        self.l = min(int(l), self.dim)
        W, kk = UniformSampling(k, self.n_objs)()
        self.k = int(kk)
        self._w = W.to(self.device)
        self.pop_size = int(math.ceil(self.pop_size / self.k) * self.k)

    def setup(self, key):
        return State(reference_vector=self._w, **self._base_fields(key))

    def ask(self, state):
        key, k_part, k_dim, k_noise, k_mut, k_fill = rnd.split(state.key, 6)
        pop, obj = state.population, state.fitness
        n, D, M, L = pop.shape[0], self.dim, self.n_objs, self.l
        dev = pop.device
        part = torch.argmax(cos_dist(obj, state.reference_vector), dim=1)
        counts = torch.bincount(part, minlength=self.k).tolist()          # the one host read of the step
        big = [k for k in range(self.k) if counts[k] >= 2 * M]
        parts = []
        if len(big) < self.k:
            # sub-populations too small for a model pass their members on unchanged
            is_big = torch.zeros(self.k, dtype=torch.bool, device=dev)
            if big:
                is_big[torch.tensor(big, device=dev)] = True
            parts.append(pop[~is_big[part]])
        groups = bool(big)
        if groups:
            # one group per (sub-population, objective): a random sample of ⌊n_k / M⌋ members, all groups
            # drawn at once — random keys, non-members pushed behind, one batched sort
            kb = torch.tensor(big, device=dev)
            gk = kb.repeat_interleave(M)                                                   # (G,)
            m_of = torch.arange(M, device=dev).repeat(len(big))                            # (G,)
            share_host = [counts[k] // M for k in big for _ in range(M)]
            G, P = len(share_host), max(share_host)
            share = torch.tensor(share_host, device=dev)
            member = part[None, :] == gk[:, None]                                          # (G, n)
            keys = torch.where(member, rnd.bits(k_part, (G, n)).to(torch.int64), torch.full((), 1 << 40, device=dev))
            order = torch.argsort(keys, dim=1)[:, :P]                                      # (G, P)
            mask = torch.arange(P, device=dev)[None, :] < share[:, None]
            idx = torch.where(mask, order, torch.zeros_like(order))
            # extremes per sub-population as masked reductions (k is small): atomic min / max of all rows onto k
            # addresses serialise
            own = (part[None, :] == torch.arange(self.k, device=dev)[:, None])[:, :, None]  # (k, n, 1)
            inf = torch.full((), float("inf"), device=dev)
            lo = torch.where(own, obj[None], inf).amin(1)                                   # (k, M)
            hi = torch.where(own, obj[None], -inf).amax(1)
            f_lo, f_hi = 1.5 * lo - 0.5 * hi, 1.5 * hi - 0.5 * lo                          # (k, M)
            a, b = f_lo[gk, m_of], f_hi[gk, m_of]                                          # (G,)
            t = torch.arange(P, device=dev)[None, :] / (share[:, None] - 1).clamp_min(1)
            grid = torch.where(mask, a[:, None] + t * (b - a)[:, None], torch.zeros((), device=dev))
            x = obj[idx, m_of[:, None]]                                                    # (G,P)
            dims = torch.argsort(rnd.bits(k_dim, (G, D)), dim=1)[:, :L]                    # (G,L)
            base = pop[idx]                                                                # (G,P,D)
            y = torch.gather(base, 2, dims[:, None, :].expand(G, P, L))
            mean, std = inverse_models(x, y, mask, grid)
            sampled = mean + rnd.normal(k_noise, (G, P, L)) * std
            base = base.scatter(2, dims[:, None, :].expand(G, P, L), sampled)
            parts.append(base[mask])
        off = torch.cat(parts, dim=0)
        have = off.shape[0]
        if have < n:
            off = torch.cat([off, pop[rnd.randint(k_fill, (n - have,), 0, n)]], dim=0)
        elif have > n:
            off = off[rnd.permutation(k_fill, have)[:n]]
        off = torch.clamp(off, self.lb, self.ub)
        nxt = torch.clamp(self.mutation(k_mut, off), self.lb, self.ub)
        return nxt, state.replace(next_generation=nxt, key=key)

    def tell(self, state, fitness):
        pop, obj = self._merged(state, fitness)
        keep = rank_crowding_truncate(pop, obj, self.pop_size)
        return state.replace(population=pop[keep], fitness=obj[keep])
