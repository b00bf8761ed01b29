"""Matrix-adaptation ES and AMaLGaM, built on the CMA-ES step-size / path machinery.

Reference: ``es_variants/ma_es.py:24-127`` and ``es_variants/amalgam.py:22-133``."""
from __future__ import annotations

import torch

from ....core import random as rnd
from ....core.state import State
from ....ops import cma as _cma
from ....ops import es as _es
from ....ops import sort as _sort
from .cma_es import CMAES


class MAES(CMAES):
    """MA-ES (Beyer & Sendhoff 2017): samples ``m + σ z Mᵀ`` and adapts the transformation
    matrix ``M`` directly; no covariance matrix, no eigendecomposition.  The update keeps the
    reference's form ``M ← (1−c₁−cμ)M + c₁(pc pcᵀ + …M) + cμ Yᵀ diag(w) Y``."""

    def setup(self, key):
        dev = self.center_init.device
        return State(pc=torch.zeros(self.dim, device=dev), ps=torch.zeros(self.dim, device=dev),
                     M=torch.eye(self.dim, device=dev), count_iter=0, mean=self.center_init,
                     count_iter_dev=torch.zeros((), device=dev),
                     sigma=torch.tensor(self.init_stdev, device=dev), key=key.to(dev),
                     population=torch.empty((self.pop_size, self.dim), device=dev))

    def graph_signature(self, state):
        return ()

    def graph_host_counters(self):
        return ("count_iter",)                 # only counts; the device work reads count_iter_dev

    def ask(self, state):
        key, sample_key = rnd.split(state.key)
        ones = torch.ones(self.dim, device=state.M.device)
        population = _cma.sample(sample_key, state.mean, state.sigma, state.M, ones, self.pop_size, self.precision)
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=state.count_iter_dev + 1, key=key)

    # Row-sharded: a rank samples its rows of the population (the counter-based generator addresses rows by their
    # index in the whole population), reduces its share of the selected steps, and one all-reduce of dim + dim² floats
    # completes the mean shift and the rank-μ term.
    def ask_sharded(self, state, rank: int, world: int):
        if self.pop_size % world:
            raise ValueError(f"pop_size {self.pop_size} is not divisible by the number of ranks {world}")
        local = self.pop_size // world
        key, sample_key = rnd.split(state.key)
        ones = torch.ones(self.dim, device=state.M.device)
        population = _cma.sample(sample_key, state.mean, state.sigma, state.M, ones, local, self.precision,
                                 row_offset=rank * local)
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=state.count_iter_dev + 1, key=key)

    def tell_sharded(self, state, fitness, rank: int, world: int):
        from ....core import distributed as dist

        local = self.pop_size // world
        _, ranking = _sort.argsort_with_rank(fitness)
        padded = torch.cat([self.weights, self.weights.new_zeros(self.pop_size - self.mu)])
        w_local = padded[ranking[rank * local:(rank + 1) * local]]          # 0 outside the best μ
        d = state.population - state.mean
        y = d / state.sigma
        gram = (y * w_local[:, None]).T @ y
        packed = dist.all_reduce(torch.cat([w_local @ d, gram.reshape(-1)]))
        mean = state.mean + self.cm * packed[: self.dim]
        ps, pc, hsig, sigma = self._paths(state, mean)
        alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
        M = alpha * state.M + self.c1 * torch.outer(pc, pc) + self.cmu * packed[self.dim:].reshape(self.dim, self.dim)
        return state.replace(mean=mean, ps=ps, pc=pc, M=M, sigma=sigma)

    def _whiten(self, state, delta):
        return state.M @ delta

    def _new_M(self, state, order, pc, hsig):
        alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
        return _cma.covariance_update(state.M, state.population, order, state.mean, state.sigma, self.weights, pc,
                                      alpha, self.c1, self.cmu, self.precision)

    def tell(self, state, fitness):
        order = _sort.argsort(fitness)
        elite = state.population[order[: self.mu]]
        mean = state.mean + self.cm * (self.weights @ (elite - state.mean))
        ps, pc, hsig, sigma = self._paths(state, mean)
        return state.replace(mean=mean, ps=ps, pc=pc, M=self._new_M(state, order, pc, hsig), sigma=sigma)


class LMMAES(MAES):
    """Limited-memory variant: the rank-μ term is built from a ring buffer of the last
    ``memory_size`` selected steps."""

    ask_sharded = None           # the step memory is built from the whole selected set in rank order
    tell_sharded = None

    def __init__(self, *args, memory_size=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.memory_size = memory_size or self.weights.numel()

    def setup(self, key):
        s = super().setup(key)
        return s.replace(y_memory=torch.zeros((self.memory_size, self.dim), device=self.center_init.device))

    def tell(self, state, fitness):
        order = _sort.argsort(fitness)
        elite = state.population[order[: self.mu]]
        mean = state.mean + self.cm * (self.weights @ (elite - state.mean))
        ps, pc, hsig, sigma = self._paths(state, mean)
        y = (elite - mean) / sigma
        y_memory = torch.cat([state.y_memory[self.mu:], y], dim=0)[-self.memory_size:]
        w = self.weights
        if w.shape[0] != y_memory.shape[0]:
            w = torch.nn.functional.pad(w, (0, y_memory.shape[0] - w.shape[0])) if w.shape[0] < y_memory.shape[0] \
                else w[: y_memory.shape[0]]
        M = ((1 - self.c1 - self.cmu) * state.M
             + self.c1 * (torch.outer(state.pc, state.pc) + (1 - hsig) * self.cc * (2 - self.cc) * state.M)
             + self.cmu * (w @ y_memory))
        return state.replace(mean=mean, ps=ps, pc=pc, M=M, sigma=sigma, y_memory=y_memory)


class AMaLGaM(CMAES):
    """CMA-ES with two ingredients of AMaLGaM (Bosman et al. 2013): the anticipated mean shift
    (the mean is pushed a further ``0.05·σ·pc`` along the evolution path) and adaptive variance
    scaling — the updated covariance is enlarged when the best sample lies more than one
    standard deviation from the mean (standard-deviation ratio ``SDR > 1``) and shrunk
    otherwise.  ``SDR = ‖C^{-1/2}(x_best − m)/σ‖ / √n``.

    The reference's version (``amalgam.py:22-68``) scales the *previous* covariance and never
    uses the rank-μ update it computes, and divides the already normalised step by σ a second
    time; both are treated as defects and not reproduced."""

    ams_alpha = 0.05
    avs_beta = 0.05
    ask_sharded = None
    tell_sharded = None

    def _scale(self, sdr, C):
        up = (1 + self.avs_beta * (sdr - 1)) * C
        down = (1 / (1 + self.avs_beta * (1 - sdr))) * C
        return torch.where(sdr > 1, up, down)

    def tell(self, state, fitness):
        order = _sort.argsort(fitness)
        elite = state.population[order[: self.mu]]
        mean = state.mean + self.cm * (self.weights @ (elite - state.mean)) + self.ams_alpha * state.sigma * state.pc
        ps, pc, hsig, sigma = self._paths(state, mean)
        alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
        C = _cma.covariance_update(state.C, state.population, order, state.mean, state.sigma, self.weights, pc,
                                   alpha, self.c1, self.cmu, self.precision)
        y0 = (elite[0] - state.mean) / state.sigma
        sdr = torch.linalg.vector_norm(state.invsqrtC @ y0) / (self.dim ** 0.5)
        C = self._scale(sdr, C)
        if state.count_iter % self.decomp_per_iter == 0:
            B, D, invsqrtC = _cma.decompose(C, state.B, self.eigh_method)
        else:
            B, D, invsqrtC = state.B, state.D, state.invsqrtC
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma, B=B, D=D, invsqrtC=invsqrtC)


class IndependentAMaLGaM(AMaLGaM):
    """Diagonal-covariance AMaLGaM."""

    def setup(self, key):
        dev = self.center_init.device
        return State(pc=torch.zeros(self.dim, device=dev), ps=torch.zeros(self.dim, device=dev),
                     C=torch.ones(self.dim, device=dev), count_iter=0, mean=self.center_init,
                     count_iter_dev=torch.zeros((), device=dev),
                     sigma=torch.tensor(self.init_stdev, device=dev), key=key.to(dev), noise_key=key.to(dev),
                     population=torch.empty((self.pop_size, self.dim), device=dev))

    def graph_signature(self, state):
        return ()

    def graph_host_counters(self):
        return ("count_iter",)                 # only counts; the device work reads count_iter_dev

    def ask(self, state):
        key, sample_key = rnd.split(state.key)
        population = _es.perturb(sample_key, state.mean, state.sigma * torch.sqrt(state.C), self.pop_size)
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=state.count_iter_dev + 1, key=key)

    def _whiten(self, state, delta):
        return delta / torch.sqrt(state.C)

    def tell(self, state, fitness):
        order = _sort.argsort(fitness)
        elite = state.population[order[: self.mu]]
        mean = state.mean + self.cm * (self.weights @ (elite - state.mean)) + self.ams_alpha * state.sigma * state.pc
        ps, pc, hsig, sigma = self._paths(state, mean)
        y = (elite - state.mean) / state.sigma
        C = ((1 - self.c1 - self.cmu) * state.C
             + self.c1 * (pc * pc + (1 - hsig) * self.cc * (2 - self.cc) * state.C)
             + self.cmu * (self.weights @ (y * y)))
        C = self._scale(y[0].abs() / torch.sqrt(state.C), C)
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma)
