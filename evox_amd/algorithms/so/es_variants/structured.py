"""CR-FM-NES, RM-ES and LES.

Reference: ``es_variants/cr_fm_nes.py:43-222``, ``es_variants/rmes.py:21-202``,
``es_variants/les.py:191-286``."""
from __future__ import annotations

import math
from typing import Optional

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.state import State
from ....ops import cma as _cma
from ....ops import es as _es
from ....ops import sort as _sort
from .cma_es import CMAES


def get_h_inv(dim: int) -> float:
    """Root of ``(1 + a²) e^{a²/2} / 0.24 − 10 − dim`` by damped Newton steps (host scalar)."""
    dim = min(dim, 2000)
    f = lambda a: ((1.0 + a * a) * math.exp(a * a / 2.0) / 0.24) - 10.0 - dim
    fp = lambda a: (1.0 / 0.24) * a * math.exp(a * a / 2.0) * (3.0 + a * a)
    h = 1.0
    while abs(f(h)) > 1e-10:
        h = h - 0.5 * (f(h) / fp(h))
    return h


class CR_FM_NES(Algorithm):
    """Cost-reduced fast-moving natural evolution strategy (Nomura & Ono 2022): covariance
    restricted to ``D (I + v vᵀ) D``, so a generation costs O(pop · dim).  Matrices are kept
    in the paper's column layout (dim × pop)."""

    def __init__(self, pop_size: int, center_init, sigma: float = 0.03):
        super().__init__()
        if pop_size % 2:
            raise ValueError("Population size must be even")
        self.center_init = torch.as_tensor(center_init, dtype=torch.float32)
        n = self.num_dims = self.dim = self.center_init.shape[0]
        lam = self.popsize = self.pop_size = pop_size
        self.sigma = sigma
        w_hat = (math.log(lam / 2 + 1) - torch.log(torch.arange(1, lam + 1, dtype=torch.float64))).clamp_min(0)
        w_rank = w_hat / w_hat.sum() - 1.0 / lam
        self.w_rank_hat, self.w_rank = w_hat.float(), w_rank.float()
        mueff = float(1 / ((w_rank + 1 / lam) ** 2).sum())
        self.mu_eff = mueff
        self.c_s = (mueff + 2.0) / (n + mueff + 5.0)
        self.c_c = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
        c1_cma = 2.0 / ((n + 1.3) ** 2 + mueff)
        self.chi_N = math.sqrt(n) * (1.0 - 1.0 / (4.0 * n) + 1.0 / (21.0 * n * n))
        self.h_inv = get_h_inv(n)
        self.alpha_dist = self.h_inv * min(1.0, math.sqrt(lam / n))
        self.lrate_mean = 1.0
        self.lrate_move_sigma = 1.0
        self.lrate_stag_sigma = math.tanh((0.024 * lam + 0.7 * n + 20.0) / (n + 12.0))
        self.lrate_conv_sigma = 2.0 * math.tanh((0.025 * lam + 0.75 * n + 10.0) / (n + 4.0))
        self.c1 = c1_cma * (n - 5) / 6
        self.lrate_B = math.tanh((min(0.02 * lam, 3 * math.log(n)) + 5) / (0.23 * n + 25))

    def setup(self, key):
        dev = self.center_init.device
        key, k_v = rnd.split(key.to(dev))
        n = self.num_dims
        return State(key=key, noise_key=key, sigma=torch.tensor(float(self.sigma), device=dev), center=self.center_init,
                     v=rnd.normal(k_v, (n, 1)) / math.sqrt(n), D=torch.ones((n, 1), device=dev),
                     p_sigma=torch.zeros((n, 1), device=dev), p_c=torch.zeros((n, 1), device=dev),
                     w_rank_hat=self.w_rank_hat.to(dev)[:, None], w_rank=self.w_rank.to(dev)[:, None])

    def _zy(self, state, noise_key):
        h = self.popsize // 2
        zp = _es.noise_rows(noise_key, h, self.num_dims)
        z = torch.cat([zp, -zp], dim=0).T                                   # (dim, pop)
        normv2 = (state.v * state.v).sum()
        vbar = state.v / torch.sqrt(normv2)
        y = z + (torch.sqrt(1 + normv2) - 1) * (vbar @ (vbar.T @ z))
        return z, y

    def ask(self, state):
        key, noise_key = rnd.split(state.key)
        _, y = self._zy(state, noise_key)
        x = (state.center[:, None] + state.sigma * y * state.D).T
        return x, state.replace(key=key, noise_key=noise_key)

    def tell(self, state, fitness):
        n, lam = self.num_dims, self.popsize
        ranks = torch.argsort(fitness, stable=True)
        z, y = self._zy(state, state.noise_key)
        z, y = z[:, ranks], y[:, ranks]
        x = state.center[:, None] + state.sigma * y * state.D
        p_sigma = (1 - self.c_s) * state.p_sigma + math.sqrt(self.c_s * (2.0 - self.c_s) * self.mu_eff) * (z @ state.w_rank)
        ps_norm = torch.linalg.vector_norm(p_sigma)
        w_tmp = state.w_rank_hat * torch.exp(self.alpha_dist * torch.linalg.vector_norm(z, dim=0))[:, None]
        weights_dist = w_tmp / w_tmp.sum() - 1.0 / lam
        moving = ps_norm >= self.chi_N
        weights = torch.where(moving, weights_dist, state.w_rank)
        lrate_sigma = torch.where(moving, float(self.lrate_move_sigma), float(self.lrate_stag_sigma))
        lrate_sigma = torch.where(ps_norm >= 0.1 * self.chi_N, lrate_sigma, float(self.lrate_conv_sigma))
        wxm = (x - state.center[:, None]) @ weights
        p_c = (1.0 - self.c_c) * state.p_c + math.sqrt(self.c_c * (2.0 - self.c_c) * self.mu_eff) * wxm / state.sigma
        center = state.center + self.lrate_mean * wxm[:, 0]
        normv2 = (state.v * state.v).sum()
        normv = torch.sqrt(normv2)
        normv4 = normv2 * normv2
        vbar = state.v / normv
        exY = torch.cat([y, p_c / state.D], dim=1)
        yy = exY * exY
        ip_yvbar = vbar.T @ exY
        yvbar = exY * vbar
        gammav = 1.0 + normv2
        vbarbar = vbar * vbar
        alphavd = torch.clamp(torch.sqrt(normv4 + (2 * gammav - torch.sqrt(gammav)) / vbarbar.max()) / (2 + normv2), max=1.0)
        t = exY * ip_yvbar - vbar * (ip_yvbar ** 2 + gammav) / 2
        b = -(1 - alphavd ** 2) * normv4 / gammav + 2 * alphavd ** 2
        H = 2.0 - (b + 2 * alphavd ** 2) * vbarbar
        invH = 1.0 / H
        s1 = yy - normv2 / gammav * (yvbar * ip_yvbar) - 1.0
        ip_vbart = vbar.T @ t
        s2 = s1 - alphavd / gammav * ((2 + normv2) * (t * vbar) - normv2 * (vbarbar @ ip_vbart))
        invHvbarbar = invH * vbarbar
        ip_s2 = invHvbarbar.T @ s2
        s = (s2 * invH) - b / (1 + b * (vbarbar.T @ invHvbarbar)) * (invHvbarbar @ ip_s2)
        ip_svbarbar = vbarbar.T @ s
        t = t - alphavd * ((2 + normv2) * (s * vbar) - vbar @ ip_svbarbar)
        exw = torch.cat([self.lrate_B * weights, torch.full((1, 1), float(self.c1), device=z.device)], dim=0)
        v = state.v + (t @ exw) / normv
        D = state.D + (s @ exw) * state.D
        nthroot = torch.exp(torch.log(D).sum() / n + torch.log(1 + (v * v).sum()) / (2 * n))
        D = D / nthroot
        G_s = ((z * z - 1.0) @ weights).sum() / n
        sigma = state.sigma * torch.exp(lrate_sigma / 2 * G_s)
        return state.replace(p_sigma=p_sigma, center=center, p_c=p_c, v=v, D=D, sigma=sigma)


class RMES(CMAES):
    """CMA-ES augmented with the bookkeeping of the rank-m ES (Li & Zhang 2018) as in the
    reference: a sparse-plus-low-rank rank-one term, a memory ``P`` of past σ-paths kept at
    least ``t_uncorr`` generations apart and a rank-based success rate.  The mean follows the
    usual weighted recombination (with the reference's defaults, ``mean_decay = 0`` freezes the
    mean, ``rmes.py:110-115``)."""

    def __init__(self, center_init, init_stdev, pop_size=None, recombination_weights=None, cm=1, elite_ratio=0.5,
                 memory_size=10, mean_decay=0, sparse_threshold=0.01, t_uncorr=10, **kwargs):
        super().__init__(center_init, init_stdev, pop_size, recombination_weights, cm, **kwargs)
        self.elite_ratio = elite_ratio
        self.elite_popsize = max(1, int(self.pop_size * elite_ratio))
        self.memory_size, self.mean_decay = memory_size, mean_decay
        self.sparse_threshold, self.t_uncorr = sparse_threshold, t_uncorr
        self.c_sigma = (self.mueff + 2) / (self.dim + self.mueff + 5)
        self.q_star, self.c_s = 0.5, 0.3

    ask_sharded = None
    tell_sharded = None

    def setup(self, key):
        dev = self.center_init.device
        s = super().setup(key)
        return s.replace(P=torch.zeros((self.dim, self.memory_size), device=dev),
                         t_gap=torch.zeros(self.memory_size, device=dev), s_rank_rate=torch.zeros((), device=dev),
                         fitness_archive=torch.full((self.pop_size,), 1e20, device=dev),
                         p_sigma=torch.zeros(self.dim, device=dev))

    def _update_P(self, P, p_sigma, t_gap, gen):
        """``gen``: the generation number as a 0-d device tensor.  The oldest path leaves the memory when all paths are
        at least ``t_uncorr`` generations apart (or the memory is still filling), otherwise the newer one of the closest
        pair — decided on the device (a column gather), so the step has no host read."""
        m = self.memory_size
        gaps = t_gap[1:] - t_gap[:-1]
        push = (gaps.min() > self.t_uncorr) | (gen < m)
        drop = torch.where(push, torch.zeros((), dtype=torch.int64, device=P.device), torch.argmin(gaps))
        idx = torch.arange(m - 1, device=P.device)
        keep = idx + (idx >= drop).to(idx.dtype)
        P = torch.cat([P[:, keep], p_sigma[:, None]], dim=1)
        t_gap = torch.cat([t_gap[keep], gen.reshape(1).to(t_gap.dtype)])
        return P, t_gap

    def _rank_success(self, state, sorted_fitness):
        mu = self.weights.shape[0]
        n = sorted_fitness.shape[0]
        both = torch.cat([sorted_fitness, state.fitness_archive])
        ranks = torch.empty(2 * n, device=both.device)
        ranks[torch.argsort(both, stable=True)] = torch.arange(2 * n, device=both.device, dtype=ranks.dtype)
        cur = torch.sort(ranks[:n]).values[:mu]
        last = torch.sort(ranks[n:]).values[:mu]
        q = (self.weights * (last - cur)).sum() / mu
        return (1 - self.c_s) * state.s_rank_rate + self.c_s * (q - self.q_star)

    def tell(self, state, fitness):
        order = _sort.argsort(fitness)
        elite = state.population[order[: self.mu]]
        mean = state.mean + self.cm * (self.weights @ (elite - state.mean))
        ps, pc, hsig, sigma = self._paths(state, mean)
        y = (elite - state.mean) / state.sigma
        cross = torch.outer(y.mean(0), pc)
        rank_one = torch.where(cross.abs() > self.sparse_threshold, cross, torch.zeros_like(cross)) + torch.outer(pc, pc)
        C = ((1 - self.c1 - self.cmu) * state.C
             + self.c1 * (rank_one + (1 - hsig) * self.cc * (2 - self.cc) * state.C)
             + self.cmu * ((y.T * self.weights) @ y))
        C = 0.5 * (C + C.T)
        if state.count_iter % self.decomp_per_iter == 0:
            B, D, invsqrtC = _cma.decompose(C, state.B, self.eigh_method)
        else:
            B, D, invsqrtC = state.B, state.D, state.invsqrtC
        p_sigma = (1 - self.c_sigma) * state.p_sigma + math.sqrt(self.c_sigma * (2 - self.c_sigma) * self.mueff) * (
            mean - state.mean) / state.sigma
        P, t_gap = self._update_P(state.P, state.p_sigma, state.t_gap, state.count_iter_dev)
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma, B=B, D=D, invsqrtC=invsqrtC, p_sigma=p_sigma,
                             P=P, t_gap=t_gap, s_rank_rate=self._rank_success(state, fitness[order]),
                             fitness_archive=fitness[order])


# -- LES ---------------------------------------------------------------------------------------
_TIMESCALES = (1, 3, 10, 30, 50, 100, 250, 500, 750, 1000, 1250, 1500, 2000)


def tanh_timestamp(x) -> torch.Tensor:
    """``x``: generation number — a Python number, or a 0-d device tensor (the embedding is then computed on its device)."""
    if isinstance(x, torch.Tensor):
        t = torch.tensor(_TIMESCALES, dtype=torch.float32, device=x.device) if x.device.type == "cpu" else _timescales_on(x.device)
        return torch.tanh(x.to(torch.float32) / t - 1.0)
    t = torch.tensor(_TIMESCALES, dtype=torch.float32)
    return torch.tanh(float(x) / t - 1.0)


_TS_CACHE = {}


def _timescales_on(device) -> torch.Tensor:
    t = _TS_CACHE.get(device)
    if t is None:
        t = _TS_CACHE[device] = torch.tensor(_TIMESCALES, dtype=torch.float32, device=device)
    return t


def _dense(p, x):
    return x @ p["kernel"] + p["bias"]


def init_les_params(key, att_hidden_dims: int = 8, mlp_hidden_dims: int = 8, n_features: int = 3, n_paths: int = 3,
                    device=None):
    """Randomly initialised parameter tree with the layout of the LES networks:
    ``{"recomb_weights": {keys, queries, values}, "lrate_modulation": {hidden, mean, sigma}}``,
    each a ``{"kernel", "bias"}`` dense layer."""
    ks = rnd.split(key, 6)

    def dense(k, i, o):
        return {"kernel": (rnd.normal(k, (i, o)) / math.sqrt(i)).to(device) if device else rnd.normal(k, (i, o)) / math.sqrt(i),
                "bias": torch.zeros(o, device=device)}

    mlp_in = 2 * n_paths + len(_TIMESCALES)
    return {"recomb_weights": {"keys": dense(ks[0], n_features, att_hidden_dims),
                               "queries": dense(ks[1], n_features, att_hidden_dims),
                               "values": dense(ks[2], n_features, 1)},
            "lrate_modulation": {"hidden": dense(ks[3], mlp_in, mlp_hidden_dims), "mean": dense(ks[4], mlp_hidden_dims, 1),
                                 "sigma": dense(ks[5], mlp_hidden_dims, 1)}}


class LES(Algorithm):
    """Learned evolution strategy (Lange et al. 2023): recombination weights come from a small
    self-attention network over fitness features, learning rates from an MLP over evolution
    paths at three time scales.

    The pretrained checkpoint the reference expects (``2023_03_les_v1.pkl``) is not part of the
    reference tree, so parameters must be supplied: ``net_params`` (tree as produced by
    :func:`init_les_params`) or ``net_ckpt_path`` (a safetensors / ``torch.save`` file of such a
    tree, loaded with ``weights_only=True``).  Without either, randomly initialised networks
    are used and a warning is issued."""

    def __init__(self, pop_size: int, center_init, sigma_init: float = 0.1, mean_decay: float = 0.0, net_params=None,
                 net_ckpt_path: Optional[str] = None):
        super().__init__()
        self.center_init = torch.as_tensor(center_init, dtype=torch.float32)
        self.num_dims = self.dim = self.center_init.shape[0]
        self.popsize = self.pop_size = pop_size
        self.sigma_init = sigma_init
        self.timescales = torch.tensor([0.1, 0.5, 0.9])
        dev = self.center_init.device
        if net_ckpt_path is not None:
            net_params = torch.load(net_ckpt_path, weights_only=True)
        if net_params is None:
            import warnings

            warnings.warn("LES: no pretrained parameters available, using randomly initialised networks")
            net_params = init_les_params(rnd.PRNGKey(2023))
        from ....core.tree import tree_map

        self.les_net_params = tree_map(lambda t: torch.as_tensor(t, dtype=torch.float32).to(dev), net_params)

    def setup(self, key):
        dev = self.center_init.device
        key = key.to(dev)
        zeros = torch.zeros((self.num_dims, 3), device=dev)
        return State(key=key, noise_key=key, sigma=torch.full((self.num_dims,), float(self.sigma_init), device=dev),
                     mean=self.center_init, path_c=zeros, path_sigma=zeros.clone(),
                     best_fitness=torch.tensor(torch.finfo(torch.float32).max, device=dev),
                     best_member=self.center_init, gen_counter=0,
                     # the same counter on the device: the time embedding of a recorded generation follows it
                     gen_counter_dev=torch.zeros((), device=dev))

    def graph_host_counters(self):
        return ("gen_counter",)

    def ask(self, state):
        key, noise_key = rnd.split(state.key)
        x = _es.perturb(noise_key, state.mean, state.sigma, self.popsize)
        return x, state.replace(key=key, noise_key=noise_key)

    def _features(self, fitness, best):
        n = fitness.shape[0]
        better = (fitness < best).float()
        ranks = torch.empty(n, device=fitness.device)
        ranks[torch.argsort(fitness, stable=True)] = torch.arange(n, device=fitness.device, dtype=torch.float32)
        cr = ranks / (n - 1) - 0.5
        zs = (fitness - fitness.mean()) / (fitness.std(unbiased=False) + 1e-10)
        return torch.stack([better, cr, zs], dim=1)

    def tell(self, state, fitness):
        p = self.les_net_params
        dev = fitness.device
        feats = self._features(fitness, state.best_fitness)
        att = p["recomb_weights"]
        keys, queries, values = _dense(att["keys"], feats), _dense(att["queries"], feats), _dense(att["values"], feats)
        A = torch.softmax(queries @ keys.T / math.sqrt(feats.shape[0]), dim=-1)
        weights = torch.softmax((A @ values)[:, 0], dim=0)                       # (pop,)
        # x_i − mean = σ z_i, so every weighted sum over the population is a weighted noise sum
        g, s = _es.weighted_noise_sum(state.noise_key, weights, self.num_dims, v=weights)
        weight_noise = g
        weight_diff = state.sigma * g
        if self.timescales.device != dev:
            self.timescales = self.timescales.to(dev)         # once: a copy per generation would block graph capture
        ts = self.timescales
        path_c = (1 - ts) * state.path_c + (1 - ts) * weight_diff[:, None]
        path_sigma = (1 - ts) * state.path_sigma + (1 - ts) * weight_noise[:, None]
        time_embed = tanh_timestamp(state.gen_counter_dev)
        X = torch.cat([path_c, path_sigma, time_embed[None, :].expand(self.num_dims, -1)], dim=1)
        mlp = p["lrate_modulation"]
        hidden = torch.relu(_dense(mlp["hidden"], X))
        lr_mean = torch.sigmoid(_dense(mlp["mean"], hidden))[:, 0]
        lr_sigma = torch.sigmoid(_dense(mlp["sigma"], hidden))[:, 0]
        weighted_mean = state.mean + weight_diff
        weighted_sigma = torch.sqrt(state.sigma ** 2 * (s + 1.0) + 1e-10)
        mean = state.mean + lr_mean * (weighted_mean - state.mean)
        sigma = torch.clamp(state.sigma + lr_sigma * (weighted_sigma - state.sigma), min=0)
        best_fitness = torch.minimum(fitness.min(), state.best_fitness)
        return state.replace(mean=mean, sigma=sigma, path_c=path_c, path_sigma=path_sigma, best_fitness=best_fitness,
                             gen_counter=state.gen_counter + 1, gen_counter_dev=state.gen_counter_dev + 1)
