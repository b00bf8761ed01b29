"""Many equally configured CMA-ES instances advanced together.

The decomposition containers of the reference vmap one algorithm over stacked states, i.e. run one fused program for
all blocks (``src/evox/algorithms/containers/clustered_algorithm.py:10-42``, ``coevolution.py:12-136``).  Here the same
is a CMA-ES whose state carries a leading block axis: ``S`` means, step sizes, paths and ``d x d`` covariance matrices,
sampled by one batched product, updated by batched rank-μ products and factorised by one batched symmetric
eigendecomposition.  A generation costs the launches of ONE member, whatever ``S`` is.

Every block follows exactly the update of :class:`~evox_amd.algorithms.so.es_variants.cma_es.CMAES` (same constants,
same order of operations); the blocks differ only in their random draws, which come from one stream cut into ``S``
slabs.  The fitness is either shared by all blocks (``(P,)``: the clustered container, where every block is told the
fitness of the concatenated candidates) or given per block (``(S, P)``: cooperative co-evolution).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.state import State
from ....ops import cma as _cma


def _batched_eigh(C: torch.Tensor, B_prev: Optional[torch.Tensor] = None):
    """Eigenvectors and eigenvalues of a stack of symmetric matrices (upper triangles are read).  On a GPU the stack goes
    to the batched Jacobi kernel — one workgroup per matrix, all of them at once, warm-started from the previous
    eigenvectors (``ops/linalg_small.py``); with that switched off, small matrices take the host round trip."""
    C = torch.triu(C) + torch.triu(C, 1).transpose(-1, -2)
    from ....ops import linalg_small as _ls

    if _ls.device_linalg(C, C.shape[-1]):
        evals, B = _ls.eigh_small(C, B_prev, spd=True)
        return B.contiguous(), evals
    if C.is_cuda and C.shape[-1] <= _cma._host_max():
        evals, B = torch.linalg.eigh(C.cpu())
        return B.to(C.device, non_blocking=True), evals.to(C.device, non_blocking=True)
    evals, B = torch.linalg.eigh(C)
    return B, evals


class BatchedCMAES(Algorithm):
    """``num_blocks`` CMA-ES instances over ``block_dim`` variables each.

    ``ask`` returns the candidates of all blocks side by side, ``(pop_size, num_blocks · block_dim)`` — block ``s`` owns
    the columns ``[s·d, (s+1)·d)`` — or, with ``concat=False``, as ``(num_blocks, pop_size, block_dim)``.
    ``tell`` takes ``(pop_size,)`` (one fitness for all blocks) or ``(num_blocks, pop_size)``."""

    def __init__(self, center_init, init_stdev, pop_size: Optional[int] = None, recombination_weights=None, cm=1,
                 decomp_per_iter: Optional[int] = None, concat: bool = True):
        super().__init__()
        center = torch.as_tensor(center_init, dtype=torch.float32)
        if center.dim() != 2:
            raise ValueError("center_init must have shape (num_blocks, block_dim)")
        if not init_stdev > 0:
            raise ValueError("init_stdev must be positive")
        self.center_init = center
        self.init_stdev = float(init_stdev)
        self.num_blocks, self.block_dim = center.shape
        self.dim = self.num_blocks * self.block_dim
        self.cm = cm
        self.concat = bool(concat)
        d = self.block_dim
        self.pop_size = 4 + math.floor(3 * math.log(d)) if pop_size is None else int(pop_size)
        if recombination_weights is None:
            self.mu = self.pop_size // 2
            w = math.log(self.mu + 0.5) - torch.log(torch.arange(1, self.mu + 1, dtype=torch.float64))
            w = w / w.sum()
        else:
            w = torch.as_tensor(recombination_weights, dtype=torch.float64).cpu()
            if not bool((w[1:] <= w[:-1]).all()) or not bool((w > 0).all()) or abs(float(w.sum()) - 1) >= 1e-6:
                raise ValueError("recombination_weights must be positive, non-increasing and sum to 1")
            self.mu = w.shape[0]
            if self.mu > self.pop_size:
                raise ValueError("more recombination weights than individuals")
        self.weights = w.to(torch.float32).to(center.device)
        n = float(d)
        self.mueff = float(w.sum() ** 2 / (w ** 2).sum())
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)
        self.cs = (2 + self.mueff) / (n + self.mueff + 5)
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))
        self.damps = 1 + 2 * max(0.0, math.sqrt((self.mueff - 1) / (n + 1)) - 1) + self.cs
        self.chiN = n ** 0.5 * (1 - 1 / (4 * n) + 1 / (21 * n ** 2))
        if decomp_per_iter is None:
            decomp_per_iter = max(int(math.floor(1 / (self.c1 + self.cmu) / n / 10)), 1)
        self.decomp_per_iter = int(decomp_per_iter)

    @classmethod
    def of(cls, member, times: int, concat: bool = True) -> "BatchedCMAES":
        """``times`` copies of the CMA-ES ``member`` (same centre, step size, weights and constants) as one instance."""
        w = member.weights.to(torch.float64)
        new = cls(member.center_init.expand(times, member.dim).clone(), member.init_stdev, member.pop_size, w / w.sum(),
                  member.cm, member.decomp_per_iter, concat)
        # the member derived its constants from the weights in double precision: take them, and the weights, as they are
        new.weights = member.weights.clone()
        for name in ("mueff", "cc", "cs", "c1", "cmu", "damps", "chiN"):
            setattr(new, name, getattr(member, name))
        return new

    def setup(self, key):
        dev = self.center_init.device
        S, d = self.num_blocks, self.block_dim
        eye = torch.eye(d, device=dev).expand(S, d, d).contiguous()
        return State(pc=torch.zeros(S, d, device=dev), ps=torch.zeros(S, d, device=dev), B=eye, D=torch.ones(S, d, device=dev),
                     C=eye.clone(), count_eigen=0, count_iter=0, invsqrtC=eye.clone(), mean=self.center_init,
                     sigma=torch.full((S,), self.init_stdev, device=dev), key=key.to(dev),
                     population=torch.empty((S, self.pop_size, d), device=dev), count_iter_dev=torch.zeros((), device=dev))

    def graph_signature(self, state):
        return ((state.count_iter + 1) % self.decomp_per_iter == 0,)

    # Recording into a HIP graph: the generation counters live on the host (the device work reads ``count_iter_dev``),
    # and a generation that factorises is a different recording from one that does not.
    def graph_host_counters(self):
        return ("count_iter", "count_eigen")

    def graph_variant(self, state):
        return self.graph_signature(state)[0]

    # ------------------------------------------------------------------------------------
    def sample(self, state, z):
        """Candidates of standard normal draws ``z`` (S, P, d): ``m + σ · B (D ∘ z)`` per block."""
        y = torch.bmm(z * state.D[:, None, :], state.B.transpose(1, 2))
        return torch.addcmul(state.mean[:, None, :], state.sigma[:, None, None], y)

    def ask(self, state):
        key, sample_key = rnd.split(state.key)
        z = rnd.normal(sample_key, (self.num_blocks, self.pop_size, self.block_dim)).to(state.mean.device)
        pop = self.sample(state, z)
        state = state.replace(population=pop, count_iter=state.count_iter + 1, count_iter_dev=state.count_iter_dev + 1,
                              key=key)
        return (self.as_candidates(pop) if self.concat else pop), state

    def as_candidates(self, pop):
        """(S, P, d) -> (P, S·d): the blocks side by side."""
        return pop.permute(1, 0, 2).reshape(self.pop_size, self.dim)

    def tell(self, state, fitness):
        S, d, mu = self.num_blocks, self.block_dim, self.mu
        pop = state.population
        if fitness.dim() == 1:
            order = torch.argsort(fitness, stable=True)[:mu]
            elite = pop[:, order]                                             # (S, mu, d)
        else:
            order = torch.argsort(fitness, dim=1, stable=True)[:, :mu]
            elite = torch.gather(pop, 1, order[:, :, None].expand(S, mu, d))
        sigma = state.sigma
        y = (elite - state.mean[:, None, :]) / sigma[:, None, None]
        step = torch.einsum("k,skd->sd", self.weights, y)
        delta = self.cm * sigma[:, None] * step
        mean = state.mean + delta
        z = torch.bmm(state.invsqrtC, (delta / sigma[:, None])[:, :, None]).squeeze(-1)
        ps = (1 - self.cs) * state.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * z
        ps_norm = torch.linalg.vector_norm(ps, dim=1)
        denom = torch.sqrt(1 - (1 - self.cs) ** (2 * state.count_iter_dev))
        hsig = (ps_norm / denom < (1.4 + 2 / (d + 1)) * self.chiN).to(torch.float32)
        pc = (1 - self.cc) * state.pc + (hsig * math.sqrt(self.cc * (2 - self.cc) * self.mueff))[:, None] * delta / sigma[:, None]
        new_sigma = sigma * torch.exp((self.cs / self.damps) * (ps_norm / self.chiN - 1))
        alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
        rank_mu = torch.bmm((y * self.weights[None, :, None]).transpose(1, 2), y)
        C = alpha[:, None, None] * state.C + self.c1 * pc[:, :, None] * pc[:, None, :] + self.cmu * rank_mu
        if state.count_iter % self.decomp_per_iter == 0:
            B, evals = _batched_eigh(C, state.B)
            D = torch.sqrt(evals.clamp_min(1e-30))
            invsqrtC = torch.bmm(B / D[:, None, :], B.transpose(1, 2))
            count_eigen = state.count_eigen + 1
        else:
            B, D, invsqrtC, count_eigen = state.B, state.D, state.invsqrtC, state.count_eigen
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=new_sigma, B=B, D=D, invsqrtC=invsqrtC,
                             count_eigen=count_eigen)
