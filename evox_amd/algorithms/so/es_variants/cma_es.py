"""CMA-ES family: CMAES, SepCMAES, IPOPCMAES, BIPOPCMAES.

Follows Hansen, "The CMA Evolution Strategy: A Tutorial" with the parameter settings of the
reference (``src/evox/algorithms/so/es_variants/cma_es.py:32-388``).  The three heavy steps
(sampling GEMM, rank-μ covariance update, eigendecomposition) are ``evox_amd.ops.cma``; the
O(n) path/step-size algebra is a few small tensor ops.

State layout equals the reference's: ``pc, ps, B, D, C, count_eigen, count_iter, invsqrtC,
mean, sigma, key, population``.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.state import State
from ....ops import cma as _cma
from ....ops import cma_bf16 as _b16
from ....ops import sort as _sort


class CMAES(Algorithm):
    def __init__(self, center_init, init_stdev, pop_size: Optional[int] = None, recombination_weights=None, cm=1,
                 decomp_per_iter: Optional[int] = None, precision: str = "fp32", eigh_method: Optional[str] = None,
                 decomposition: str = "eigh", deterministic: bool = False, deterministic_blocks: int = 8):
        """
        center_init            initial mean, shape (dim,); its device is where the run lives.
        init_stdev             initial step size σ > 0.
        pop_size               λ (default 4 + ⌊3 ln dim⌋).
        recombination_weights  positive, non-increasing, summing to 1 (default log weights).
        decomp_per_iter        eigendecompose every this many generations (default: the
                               tutorial's max(⌊1/((c₁+cμ)·dim·10)⌋, 1)).
        precision              "fp32" or "bf16" inputs of the sampling / covariance GEMMs
                               (accumulation is always fp32).
        eigh_method            see ``evox_amd.ops.cma.decompose``.
        deterministic          row-sharded runs (``ShardedWorkflow``) give the same bits for every number of
                               processes that divides ``deterministic_blocks``: the population is handled in
                               that many fixed row blocks, whichever process owns them (sampling, selection
                               and the partial sums of the update per block; the blocks' sums are gathered and
                               added in block order).  Costs a larger message (one partial covariance per
                               block instead of one all-reduced sum) and smaller kernels.
        decomposition          "eigh": ``B``/``D`` are eigenvectors / sqrt-eigenvalues of ``C`` (the
                               tutorial's formulation).  "newton_schulz": the two matrix functions
                               the algorithm consumes, ``C^{1/2}`` (sampling) and ``C^{-1/2}``
                               (σ-path), are computed directly by a GEMM-only iteration; the state
                               then holds ``B = C^{1/2}``, ``D = 1`` so that the sampling rule
                               ``m + σ (Z∘D) Bᵀ`` and ``B D² Bᵀ = C`` keep holding.  Both draw
                               ``N(m, σ²C)`` and give the same ``ps``; only the noise-to-sample map
                               differs by an orthogonal factor.
        """
        super().__init__()
        self.center_init = torch.as_tensor(center_init, dtype=torch.float32)
        if not init_stdev > 0:
            raise ValueError("init_stdev must be positive")
        self.init_stdev = float(init_stdev)
        self.dim = self.center_init.shape[0]
        self.cm = cm
        self.precision = precision
        self.eigh_method = eigh_method
        # row-sharded runs: the same bits whatever the number of processes (see ``_tell_sharded_blocks``)
        self.deterministic = bool(deterministic)
        self.deterministic_blocks = int(deterministic_blocks)
        if decomposition not in ("eigh", "newton_schulz"):
            raise ValueError("decomposition must be 'eigh' or 'newton_schulz'")
        self.decomposition = decomposition
        self.pop_size = 4 + math.floor(3 * math.log(self.dim)) if pop_size is None else int(pop_size)
        dev = self.center_init.device
        if recombination_weights is None:
            self.mu = self.pop_size // 2
            w = math.log(self.mu + 0.5) - torch.log(torch.arange(1, self.mu + 1, dtype=torch.float64))
            w = w / w.sum()
        else:
            w = torch.as_tensor(recombination_weights, dtype=torch.float64).cpu()
            if not bool((w[1:] <= w[:-1]).all()):
                raise ValueError("recombination_weights must be non-increasing")
            if abs(float(w.sum()) - 1) >= 1e-6:
                raise ValueError("sum of recombination_weights must be 1")
            if not bool((w > 0).all()):
                raise ValueError("recombination_weights must be positive")
            self.mu = w.shape[0]
            if self.mu > self.pop_size:
                raise ValueError("more recombination weights than individuals")
        self.weights = w.to(torch.float32).to(dev)
        # non-negative weights allow the Gram form of the rank-μ term (S = sqrt(w)·Y, SᵀS)
        self._sqrt_w = torch.sqrt(self.weights) if bool((w >= 0).all()) else None
        n = float(self.dim)
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
        # bf16 operands on a GPU: the sampling product, the rank-μ update and the factorisation all run on the
        # hand-written bf16 matrix-core kernels (``EVOX_AMD_CMA_BF16=library`` keeps the vendor-library products)
        self._bf16_kernels = (precision == "bf16" and self.center_init.is_cuda and self.dim % 4 == 0
                              and self._sqrt_w is not None and os.environ.get("EVOX_AMD_CMA_BF16", "kernels") == "kernels"
                              and os.environ.get("EVOX_AMD_GEMM", "hip") != "torch")
        self._workspaces = {}
        self._factor_cache = (None, None)        # (fp32 factor tensor, its padded bf16 image)
        self._split_ahead = None                 # (key tensor, next key, sampling key) made one generation ahead
        # the whole kernel sequence of ask / tell from one native call each (``EVOX_AMD_CMA_NATIVE=0``: one call per op)
        self._native_calls = self._bf16_kernels and os.environ.get("EVOX_AMD_CMA_NATIVE", "1") != "0"
        self._prefetch_draws = os.environ.get("EVOX_AMD_CMA_PREFETCH", "0") == "1"   # measured slower: the draw kernel competes with the factorisation for CUs
        self._generations = {}
        if decomposition == "newton_schulz":
            planes = int(os.environ.get("EVOX_AMD_NS_PLANES", "2")) if self._bf16_kernels else 0
            self._ns = _cma.NewtonSchulzSqrt(bf16_copy=precision == "bf16" and not planes, planes=planes,
                                             initial_cond=1.0)        # the covariance starts as the identity
        else:
            self._ns = None
        # eager generations before a workflow first records the step into a HIP graph: the factorisation's iteration
        # control starts conservative and settles within a couple of dozen calls, one decision per call
        self.graph_settle_steps = 24 if self._ns is not None else 0

    # ------------------------------------------------------------------------------------
    def setup(self, key):
        dev = self.center_init.device
        eye = torch.eye(self.dim, device=dev)
        extra = self._ns.init_control(self.dim, dev) if self._ns is not None else {}
        return State(
            pc=torch.zeros(self.dim, device=dev), ps=torch.zeros(self.dim, device=dev), B=eye,
            D=torch.ones(self.dim, device=dev), C=eye.clone(), count_eigen=0, count_iter=0, invsqrtC=eye.clone(),
            mean=self.center_init, sigma=torch.tensor(self.init_stdev, device=dev), key=key.to(dev),
            population=torch.empty((self.pop_size, self.dim), device=dev),
            # device-side copy of count_iter: lets the fused tell kernels (and a recorded graph of
            # the step) read the generation number without a host value
            count_iter_dev=torch.zeros((), device=dev),
            # control of the Newton–Schulz factorisation (iteration count, power vector, residual records): part
            # of the state, so that a resumed or replicated run takes the same decisions as the original
            **extra)

    # -- bf16 matrix-core path ------------------------------------------------------------------
    def _workspace(self, rows: int, mu: int) -> "_b16.Workspace":
        key = (int(rows), int(mu))
        ws = self._workspaces.get(key)
        if ws is None:
            ws = self._workspaces[key] = _b16.Workspace(self.dim, rows, mu, self.center_init.device,
                                                        ns_planes=max(self._ns.planes, 2) if self._ns is not None else 2)
        return ws

    def _generation(self, ws):
        from ....ops import cma_native as _nat
        g = self._generations.get(id(ws))
        if g is None:
            g = self._generations[id(ws)] = _nat.Generation(
                ws, self._sqrt_w, (self.cs, self.cc, self.mueff, self.cm, self.damps, self.chiN, self.c1, self.cmu), self._ns)
        return g

    def _factor_bf16(self, state, ws):
        """Zero-padded bf16 image of the sampling factor (``B`` in the Newton–Schulz mode, ``B·diag(D)``
        otherwise).  The factorisation leaves one next to the fp32 factor it returns; it is looked up by the
        identity of that tensor, so a restored or older state simply has it rebuilt."""
        src, img = self._factor_cache
        if src is state.B and img is not None:
            return img
        factor = state.B if self._ns is not None else state.B * state.D
        img = _b16.padded_bf16(factor, ws.factor_buffer().zero_())
        self._factor_cache = (state.B, img)
        return img

    def _factorise(self, state, C, ws=None):
        """(B, D, invsqrtC, control fields) of the new covariance."""
        if self._ns is not None:
            ctrl = {k: getattr(state, k) for k in self._ns.CONTROL_FIELDS}
            B, invsqrtC, b16, ctrl = self._ns.step(C, ctrl, prev=(state.B, state.invsqrtC), ws=ws)
            if b16 is not None:
                self._factor_cache = (B, b16)
            return B, state.D, invsqrtC, ctrl
        B, D, invsqrtC = _cma.decompose(C, state.B, self.eigh_method)
        return B, D, invsqrtC, {}

    def graph_signature(self, state):
        """Host-side control flow of the next ``tell`` (which HIP graph variant applies)."""
        return ((state.count_iter + 1) % self.decomp_per_iter == 0,)

    def graph_host_counters(self):
        """State fields that only count on the host (a recorded step may advance them without
        replaying host code).  Valid when every generation takes the same branch, i.e. the
        factorisation runs every generation, and the device work reads ``count_iter_dev``."""
        if self.decomp_per_iter == 1:
            return ("count_iter", "count_eigen") + (("ns_calls",) if self._ns is not None else ())
        return ()

    def graph_check(self) -> bool:
        """Polled by the workflow every few dozen replays: True asks for a new recording (the
        Newton–Schulz iteration count frozen in the graph no longer reaches the tolerance)."""
        return self._ns is not None and self._ns.needs_more_iterations()

    def graph_retune(self, state):
        """Polled by the workflow every few dozen replays of a recorded step.  A recording freezes the host decisions of
        the factorisation (iteration count, scaling mode, which driver); here the record that the last replay left on the
        device is read (one small synchronising copy) and put through the same policy as in eager mode.  Returns the
        algorithm state with the new decisions when they differ — the workflow then records again — else None."""
        ns = self._ns
        if ns is None or "ns_res1" not in state:
            return None
        rec = [float(v) for v in state.ns_res1.tolist()]
        old_iters, old_safe = int(state.ns_iters), int(state.ns_safe)
        # the safe-scaling countdown does not run inside a recording: it is advanced here, by the shortest polling period
        iters, safe = ns._policy(old_iters, max(0, old_safe - 8), rec[5], rec[6] != 0.0, max(1, old_iters))
        cond = rec[1] if not rec[0] > 0.0 else min(rec[1], ns.COND_TRUST * rec[0])
        symmetric = bool(ns.planes and 0.0 < cond <= ns.SYM_COND_MAX and safe == 0)
        old_sym = bool(int(state.ns_sym))
        if iters == old_iters and (safe > 0) == (old_safe > 0) and symmetric == old_sym:
            self._retune_relax = 0
            return None
        # a new recording costs milliseconds.  Decisions that the accuracy needs (more iterations, safe scaling, leaving
        # the bf16 driver) are taken at once; decisions that only save work must be asked for by three polls in a row, so
        # that a residual hovering around the tolerance does not make the recording flap
        needed = iters > old_iters or (safe > 0 and old_safe == 0) or (old_sym and not symmetric)
        if not needed:
            self._retune_relax = getattr(self, "_retune_relax", 0) + 1
            if self._retune_relax < 3:
                return None
        self._retune_relax = 0
        return state.replace(ns_iters=iters, ns_safe=safe, ns_sym=int(symmetric))

    def _sample_scale(self, state):
        # the Newton–Schulz mode stores the whole factor C^{1/2} in B and leaves D at ones
        return None if self._ns is not None else state.D

    def _keys(self, state):
        """``(next key, sampling key)`` of this generation: the split made ahead by the previous ``ask`` when it
        was made for this very key tensor (see ``_sample``), a fresh split otherwise — the same values either way."""
        ahead, self._split_ahead = self._split_ahead, None
        if ahead is not None and ahead[0] is state.key:
            return ahead[1], ahead[2]
        return rnd.split(state.key)

    def _sample(self, state, key, sample_key, rows: int, row_offset: int = 0, mu: Optional[int] = None):
        if self._bf16_kernels and state.mean.is_cuda:
            ws = self._workspace(rows, self.mu if mu is None else mu)
            pop = _b16.sample(sample_key, state.mean, state.sigma, self._factor_bf16(state, ws), rows, ws, row_offset)
            if not torch.cuda.is_current_stream_capturing() and os.environ.get("EVOX_AMD_CMA_PREFETCH", "0") == "1":
                # the Gaussian draws of the next generation depend on the key alone: they are produced on a side
                # stream while this generation is evaluated and told
                k2, s2 = rnd.split(key)
                self._split_ahead = (key, k2, s2)
                ws.prefetch_normal(s2, rows, self.dim, row_offset)
            return pop
        return _cma.sample(sample_key, state.mean, state.sigma, state.B, self._sample_scale(state), rows, self.precision,
                           row_offset=row_offset)

    def ask(self, state):
        if self._native_calls and state.mean.is_cuda:
            ws = self._workspace(self.pop_size, self.mu)
            population, key, count = self._generation(ws).ask(state.key, self._factor_bf16(state, ws), state.mean,
                                                              state.sigma, self.pop_size, 0, self._prefetch_draws,
                                                              count=state.count_iter_dev)
        else:
            key, sample_key = self._keys(state)
            population = self._sample(state, key, sample_key, self.pop_size)
            count = state.count_iter_dev + 1
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=count, key=key)

    # -- pieces of tell ------------------------------------------------------------------
    def _paths(self, state, mean, order=None):
        delta = mean - state.mean
        z = self._whiten(state, delta) / state.sigma
        ps = (1 - self.cs) * state.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * z
        ps_norm = torch.linalg.vector_norm(ps)
        try:
            # device-side generation counter: the bias correction then follows the counter when the step is
            # replayed from a recorded graph (a Python integer would be frozen into the recording)
            denom = torch.sqrt(1 - (1 - self.cs) ** (2 * state.count_iter_dev))
        except AttributeError:
            denom = math.sqrt(1 - (1 - self.cs) ** (2 * state.count_iter))
        hsig = (ps_norm / denom < (1.4 + 2 / (self.dim + 1)) * self.chiN).to(torch.float32)
        pc = (1 - self.cc) * state.pc + hsig * math.sqrt(self.cc * (2 - self.cc) * self.mueff) * delta / state.sigma
        sigma = state.sigma * torch.exp((self.cs / self.damps) * (ps_norm / self.chiN - 1))
        return ps, pc, hsig, sigma

    def _whiten(self, state, delta):
        if self._bf16_kernels and delta.is_cuda and state.invsqrtC.dim() == 2:
            return _b16.matvec(state.invsqrtC, delta)
        return state.invsqrtC @ delta

    def _tell_native(self, state, fitness, reduced=None, ws=None):
        """``tell`` as one native call (same kernels and arithmetic as the op-by-op path below).  ``reduced``: the
        row-sharded form, entered after the all-reduce with ``(step_sum, gram, overflow)`` (``ops/cma_native.py``)."""
        ws = self._workspace(self.pop_size, self.mu) if ws is None else ws
        gen = self._generation(ws)
        factor = state.count_iter % self.decomp_per_iter == 0
        ns = self._ns
        plan = None
        if factor and ns is not None:
            capturing = torch.cuda.is_current_stream_capturing()
            ctrl = {k: getattr(state, k) for k in ns.CONTROL_FIELDS}
            plan = ns.plan(ctrl, capturing, True, True)
        fitness = fitness.contiguous() if fitness is not None else None
        mean, ps, pc, sigma, C, B, inv, b16, vec_o, rec = gen.tell(state, fitness, factor and ns is not None, plan, reduced)
        extra = {}
        if not factor:
            B, D, inv, count_eigen = state.B, state.D, state.invsqrtC, state.count_eigen
        elif ns is not None:
            D, count_eigen = state.D, state.count_eigen + 1
            if b16 is not None:
                self._factor_cache = (B, b16)
            if capturing:
                ns._graph_scal = gen.scal[9:16]
            extra = ns.commit(ctrl, plan, rec, vec_o, capturing)
        else:
            B, D, inv, extra = self._factorise(state, C, ws)
            count_eigen = state.count_eigen + 1
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma, B=B, D=D, invsqrtC=inv, count_eigen=count_eigen,
                             **extra)

    def tell(self, state, fitness):
        if (self._native_calls and state.population.is_cuda and fitness.dtype == torch.float32
                and fitness.shape[0] == self.pop_size <= 16384):
            return self._tell_native(state, fitness)
        order = _sort.argsort(fitness)
        ws = None
        if self._bf16_kernels and state.population.is_cuda:
            # transposed bf16 factor of the selected rows, fused path kernel, upper-triangular rank-μ product
            from ....ops import gemm as _gemm
            ws = self._workspace(self.pop_size, self.mu)
            step = _b16.elite_factor_t(state.population, order, state.mean, state.sigma, self._sqrt_w, ws)
            mean, ps, pc, sigma, alpha, _ = _gemm.cma_paths(
                state.mean, step, self._whiten(state, step), state.ps, state.pc, state.sigma, state.count_iter_dev,
                (self.cs, self.cc, self.mueff, self.cm, self.damps, self.chiN, self.c1, self.cmu))
            C = _b16.covariance(state.C, pc, alpha, self.c1, self.cmu, ws)
        elif state.population.is_cuda and self._sqrt_w is not None and _cma.use_factor_form():
            # one pass builds S = sqrt(w)·(X_sel − m)/σ; the mean shift is σ·Sᵀsqrt(w) and the
            # rank-μ term SᵀS
            S, step = _cma.elite_factor(state.population, order, state.mean, state.sigma, self._sqrt_w, self.precision)
            from ....ops import gemm as _gemm
            mean, ps, pc, sigma, alpha, _ = _gemm.cma_paths(
                state.mean, step, self._whiten(state, step), state.ps, state.pc, state.sigma, state.count_iter_dev,
                (self.cs, self.cc, self.mueff, self.cm, self.damps, self.chiN, self.c1, self.cmu))
            C = _cma.covariance_from_factor(state.C, S, pc, alpha, self.c1, self.cmu)
        else:
            elite = state.population[order[: self.mu]]
            mean = state.mean + self.cm * (self.weights @ (elite - state.mean))
            ps, pc, hsig, sigma = self._paths(state, mean)
            alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
            C = _cma.covariance_update(state.C, state.population, order, state.mean, state.sigma, self.weights, pc,
                                       alpha, self.c1, self.cmu, self.precision)
        return self._finish_tell(state, mean, ps, pc, C, sigma, ws)

    def _finish_tell(self, state, mean, ps, pc, C, sigma, ws=None):
        extra = {}
        if state.count_iter % self.decomp_per_iter == 0:
            B, D, invsqrtC, extra = self._factorise(state, C, ws)
            count_eigen = state.count_eigen + 1
        else:
            B, D, invsqrtC, count_eigen = state.B, state.D, state.invsqrtC, state.count_eigen
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma, B=B, D=D, invsqrtC=invsqrtC,
                             count_eigen=count_eigen, **extra)


    # -- row-sharded protocol (evox_amd.parallel.ShardedWorkflow) ---------------------------
    # Every rank holds the replicated distribution parameters and only its own rows of the
    # population.  The counter-based generator makes rank r's rows exactly rows
    # [r·λ/W, (r+1)·λ/W) of the single-process sample, the fitness is all-gathered (λ floats),
    # and the mean / covariance updates are sums over individuals, so each rank reduces its
    # rows and one all-reduce of n + n² floats completes them.  The decomposition is
    # replicated work.
    def ask_sharded(self, state, rank: int, world: int):
        if self.pop_size % world:
            raise ValueError(f"pop_size {self.pop_size} is not divisible by the number of ranks {world}")
        local = self.pop_size // world
        consts = self._shard_consts(local, state.mean.device)
        if self._native_calls and self._bf16_kernels and state.mean.is_cuda:
            ws = self._workspace(local, min(consts["keep"], local))
            population, key, count = self._generation(ws).ask(state.key, self._factor_bf16(state, ws), state.mean, state.sigma,
                                                              local, rank * local, False, count=state.count_iter_dev)
        elif self.deterministic:
            # library GEMMs choose their kernels (and with them the order of summation) by shape: every block is sampled
            # by a call of its own, of the same shape on any number of processes
            blk, mine = self._det_blocks(world)
            key, sample_key = self._keys(state)
            population = torch.cat([self._sample(state, key, sample_key, blk, row_offset=rank * local + j * blk,
                                                 mu=min(self._shard_consts(blk, state.mean.device)["keep"], blk))
                                    for j in range(mine)])
            count = state.count_iter_dev + 1
        else:
            key, sample_key = self._keys(state)
            population = self._sample(state, key, sample_key, local, row_offset=rank * local, mu=min(consts["keep"], local))
            count = state.count_iter_dev + 1
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=count, key=key)

    def _det_blocks(self, world: int):
        nb = self.deterministic_blocks
        if nb <= 0 or self.pop_size % nb or nb % world:
            raise ValueError(f"deterministic=True: pop_size {self.pop_size} must be a multiple of deterministic_blocks {nb}, "
                             f"and that a multiple of the number of processes {world}")
        return self.pop_size // nb, nb // world

    def _tell_sharded_blocks(self, state, fitness, rank: int, world: int):
        """The row-sharded tell with a summation order that does not depend on the number of processes: every fixed row
        block contributes its own partial mean shift and rank-mu term (computed by calls of one shape), the partial sums
        of all blocks are all-gathered and added in block order on every process, and the rest is replicated work on
        identical inputs."""
        from ....core import distributed as dist

        blk, mine = self._det_blocks(world)
        local = self.pop_size // world
        pop = state.population
        dev, n = pop.device, self.dim
        ranking = _sort.argsort_with_rank(fitness)[1]
        native = self._native_calls and self._bf16_kernels and pop.is_cuda and self._sqrt_w is not None and n % 4 == 0
        nn = _b16.packed_upper_floats(n) if native else n * n
        length = -(-(nn + n + 4) // 4) * 4                      # rows of the message stay 16-byte aligned
        msgs = torch.zeros((mine, length), dtype=torch.float32, device=dev)
        consts = self._shard_consts(blk, dev)
        keep = min(consts["keep"], blk)
        ws = self._workspace(blk, keep) if native else None
        padded = None
        for j in range(mine):
            lo, rows, m = rank * local + j * blk, pop[j * blk:(j + 1) * blk], msgs[j]
            if native:
                from ....ops import cma_native as _nat
                order, sqrt_w = _nat.select_local(ranking, lo, blk, self.mu, keep, consts["sqrt_w_padded"], m[nn + n:nn + n + 4])
                _b16.elite_factor_t(rows, order, state.mean, state.sigma, sqrt_w, ws, out=m[nn:nn + n])
                _b16.gram_packed(ws, m[:nn])
            else:
                if padded is None:
                    padded = torch.cat([self.weights, self.weights.new_zeros(self.pop_size - self.mu)]).to(dev)
                w_b = padded[ranking[lo:lo + blk]]
                d = rows - state.mean
                y = d / state.sigma
                m[nn:nn + n] = w_b @ d
                m[:nn] = ((y * w_b[:, None]).T @ y).reshape(-1)
        full = dist.all_gather(msgs, dist.POP_AXIS_NAME, axis=0, tiled=True) if world > 1 else msgs
        packed = full[0].clone()
        for b in range(1, full.shape[0]):
            packed += full[b]
        if native:
            return self._tell_native(state, None, reduced=(packed[nn:nn + n], packed[:nn], None), ws=ws)
        step_sum, gram = packed[nn:nn + n], packed[:nn].view(n, n)
        mean = state.mean + self.cm * step_sum
        ps, pc, hsig, sigma = self._paths(state, mean)
        alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
        C = alpha * state.C + self.c1 * torch.outer(pc, pc) + self.cmu * gram
        return self._finish_tell(state, mean, ps, pc, C, sigma, None)

    def _shard_consts(self, local: int, device):
        """Per-process constants of the sharded tell (built once): sqrt of the recombination weights
        padded with zeros to the population size, and the identity permutation of the local rows."""
        key = (local, str(device))
        cache = self.__dict__.setdefault("_shard_cache", {})
        if key not in cache:
            padded = torch.cat([self.weights, self.weights.new_zeros(self.pop_size - self.mu)]).to(device)
            expect = local * self.mu / self.pop_size
            keep = int(min(local, -(-(expect + 6.0 * math.sqrt(local) + 1) // 64) * 64))
            cache[key] = {"sqrt_w_padded": torch.sqrt(padded), "ident": torch.arange(local, device=device),
                          "keep": keep, "calls": 0}
        return cache[key]

    def tell_sharded(self, state, fitness, rank: int, world: int, presorted=None):
        """``presorted`` — ``(message, own_rank)`` of :class:`~evox_amd.parallel.ShardedWorkflow`'s distributed ranking
        (every process's fitness values next to the same values sorted, and the rank of this process's rows among
        themselves) — spares the sort of the whole gathered fitness vector."""
        from ....core import distributed as dist

        if self.deterministic:
            return self._tell_sharded_blocks(state, fitness, rank, world)
        local = self.pop_size // world
        pop = state.population
        ws = None
        native = (self._native_calls and self._bf16_kernels and pop.is_cuda and self._sqrt_w is not None)
        presorted = presorted if native else None
        # up to the size the counting sort takes in one pass, ranking the local rows against all keys is the whole sort
        counted = (native and presorted is None and fitness.dtype == torch.float32 and fitness.is_contiguous()
                   and (rank * local) % 4 == 0 and fitness.shape[0] <= 16384)
        if not counted and presorted is None:
            ranking = _sort.argsort_with_rank(fitness)[1]
            my_rank = ranking[rank * local:(rank + 1) * local]
        if pop.is_cuda and self._sqrt_w is not None and (self._bf16_kernels or _cma.use_factor_form()):
            # local share of the sufficient statistics through the same kernels as the
            # single-process tell: S = sqrt(w_local)(X_local − m)/σ (zero rows for unselected
            # individuals), Gram matrix written straight into the message,
            # then ONE in-place all-reduce of n² + n + 4 floats
            from ....ops import gemm as _gemm
            consts = self._shard_consts(local, pop.device)
            keep = min(consts["keep"], local)
            # the message: rank-μ term, mean shift, four flag words.  The native path sends the rank-μ term as the
            # tile-packed upper triangle (it is symmetric: 0.56 n² floats instead of n²)
            nn = _b16.packed_upper_floats(self.dim) if native else self.dim * self.dim
            packed = torch.empty(nn + self.dim + 4, dtype=torch.float32, device=pop.device)   # Gram first: aligned base
            if native:
                # Only about μ/λ of the local rows are selected.  Rows are exchangeable between the ranks, so that number
                # is hypergeometric with a standard deviation of at most √local / 2: the selected rows are compacted
                # into `keep` slots (mean + 12 standard deviations; one launch, ``ops/cma_native.py::select_local``) and
                # only those go through the gather and the Gram product.  Exchangeability fails when the fitness has
                # massive ties (a converged run whose values have collapsed onto a few floats, a plateau): ties go to
                # the lower row index, i.e. to the lower ranks, and a rank can hold more selected rows than slots.  The
                # surplus rows (the last ones in row order — tied with rows that were kept) are then left out of the
                # update; every rank adds that fact to the message and the count is reported as a warning.
                from ....ops import cma_native as _nat
                if presorted is not None:
                    local_order, sqrt_w_local = _nat.presorted_select_local(presorted[0], world, rank, local, presorted[1],
                                                                            self.mu, keep, consts["sqrt_w_padded"],
                                                                            packed[nn + self.dim:])
                elif counted:
                    local_order, sqrt_w_local = _nat.rank_select_local(fitness, rank * local, local, self.mu, keep,
                                                                       consts["sqrt_w_padded"], packed[nn + self.dim:])
                else:
                    local_order, sqrt_w_local = _nat.select_local(ranking, rank * local, local, self.mu, keep,
                                                                  consts["sqrt_w_padded"], packed[nn + self.dim:])
            elif keep < local:
                local_order = _sort.argsort(my_rank.to(torch.float32))[:keep]
                packed[nn + self.dim:] = ((my_rank < self.mu).sum() > keep).to(torch.float32)
                sqrt_w_local = consts["sqrt_w_padded"][my_rank[local_order]]
            else:
                local_order = consts["ident"]
                packed[nn + self.dim:] = 0.0
                sqrt_w_local = consts["sqrt_w_padded"][my_rank]      # 0 for rows outside the best μ
            gram_out = packed[:nn] if native else packed[:nn].view(self.dim, self.dim)
            if self._bf16_kernels:
                ws = self._workspace(local, keep)
                _b16.elite_factor_t(pop, local_order, state.mean, state.sigma, sqrt_w_local, ws, out=packed[nn:nn + self.dim])
                if native:
                    _b16.gram_packed(ws, gram_out)
                else:
                    _b16.gram(ws, gram_out)
            else:
                S, step_sum = _cma.elite_factor(pop, local_order, state.mean, state.sigma, sqrt_w_local, self.precision)
                packed[nn:nn + self.dim] = step_sum
                _gemm.gram(S, out=gram_out)
            reduced = dist.all_reduce_begin(packed)
            if native and world > 1:
                # the next generation's Gaussian draws depend on the key alone: issued now, they run underneath the
                # all-reduce (which keeps only a few compute units busy) instead of after the tell
                self._generation(ws).draw_ahead(state.key, local, rank * local)
            packed = reduced.wait()
            overflow = packed[nn + self.dim]
            consts["calls"] += 1
            # The diagnosis without a host synchronisation: the flag word of every generation is copied into pinned host
            # memory asynchronously and looked at by the following calls — an overflow is reported (once) one or two
            # generations later, whenever the copy has landed.
            if pop.is_cuda and not torch.cuda.is_current_stream_capturing():
                seen = consts.get("overflow_host")
                if seen is None:
                    seen = consts["overflow_host"] = torch.zeros(1, dtype=torch.float32).pin_memory()
                if float(seen[0]) > 0 and not consts.get("overflow_warned"):
                    import warnings

                    consts["overflow_warned"] = True
                    warnings.warn(f"row-sharded CMA-ES: more than {keep} of the {local} local rows were selected on "
                                  f"{int(seen[0])} rank(s) (massive ties in the fitness?); the surplus rows were left out "
                                  "of the update", RuntimeWarning)
                seen.copy_(overflow.reshape(1), non_blocking=True)
            gram = packed[:nn] if native else packed[:nn].view(self.dim, self.dim)
            if native and ws is not None:
                # everything after the all-reduce — whitening, paths, covariance, factorisation — in one native call
                return self._tell_native(state, None, reduced=(packed[nn:nn + self.dim], gram, None), ws=ws)
            step_sum = packed[nn:nn + self.dim]
            mean, ps, pc, sigma, alpha, _ = _gemm.cma_paths(
                state.mean, step_sum, self._whiten(state, step_sum), state.ps, state.pc, state.sigma, state.count_iter_dev,
                (self.cs, self.cc, self.mueff, self.cm, self.damps, self.chiN, self.c1, self.cmu))
            C = _gemm.cma_covariance_base(state.C, pc, alpha, self.c1).add_(gram, alpha=float(self.cmu))
        else:
            padded = torch.cat([self.weights, self.weights.new_zeros(self.pop_size - self.mu)])
            w_local = padded[my_rank]
            step_sum = w_local @ (pop - state.mean)
            ident = torch.arange(local, device=pop.device)
            zero = torch.zeros(self.dim, device=pop.device)
            gram = _cma.covariance_update(torch.zeros_like(state.C), pop, ident, state.mean, state.sigma, w_local, zero,
                                          0.0, 0.0, 1.0, self.precision)
            packed = dist.all_reduce(torch.cat([step_sum, gram.reshape(-1)]))
            step_sum, gram = packed[: self.dim], packed[self.dim:].reshape(self.dim, self.dim)
            mean = state.mean + self.cm * step_sum
            ps, pc, hsig, sigma = self._paths(state, mean)
            alpha = (1 - self.c1 - self.cmu) + self.c1 * (1 - hsig) * self.cc * (2 - self.cc)
            C = alpha * state.C + self.c1 * torch.outer(pc, pc) + self.cmu * gram
        return self._finish_tell(state, mean, ps, pc, C, sigma, ws)

    tell_sharded.accepts_presorted = True


class SepCMAES(CMAES):
    """Separable CMA-ES: diagonal covariance, O(n) per individual, no eigendecomposition.

    ``blocked(times)``: ``times`` independent copies told the same fitness as ONE instance over the concatenated vector
    (what a decomposition container runs; the reference vmaps the member,
    ``src/evox/algorithms/containers/clustered_algorithm.py:10-42``): mean, paths and variances are simply wider, the
    step size, the norm of its path and the stall indicator exist once per block, and the learning rates stay those of
    the member's dimension."""

    blocks = 1
    block_fitness = False

    def blocked(self, times: int, shared_fitness: bool = True):
        """``shared_fitness=False``: ``tell`` takes one fitness column per block, ``(pop_size, times)`` (cooperative
        co-evolution, ``src/evox/algorithms/containers/coevolution.py:12-136``): every block ranks the population on
        its own column."""
        from ....core.module import clone_module

        if self.blocks != 1:
            raise ValueError("already a blocked instance")
        wide = clone_module(self)
        wide.center_init = self.center_init.repeat(times)
        wide.block_dim, wide.blocks, wide.dim = self.dim, int(times), self.dim * times
        wide.block_fitness = not shared_fitness
        return wide

    def _wide(self, per_block):
        """A per-block quantity spread over the block's variables."""
        return per_block if self.blocks == 1 else per_block.repeat_interleave(self.block_dim)

    def setup(self, key):
        dev = self.center_init.device
        return State(pc=torch.zeros(self.dim, device=dev), ps=torch.zeros(self.dim, device=dev),
                     C=torch.ones(self.dim, device=dev), count_iter=0, mean=self.center_init,
                     count_iter_dev=torch.zeros((), device=dev),
                     sigma=torch.full(() if self.blocks == 1 else (self.blocks,), float(self.init_stdev), device=dev),
                     key=key.to(dev), population=torch.empty((self.pop_size, self.dim), device=dev))

    def _paths(self, state, mean, order=None):
        if self.blocks == 1:
            return super()._paths(state, mean, order)
        d = self.block_dim
        sigma_w = self._wide(state.sigma)
        delta = mean - state.mean
        z = self._whiten(state, delta) / sigma_w
        ps = (1 - self.cs) * state.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * z
        ps_norm = torch.linalg.vector_norm(ps.reshape(self.blocks, d), dim=1)                      # one per block
        denom = torch.sqrt(1 - (1 - self.cs) ** (2 * state.count_iter_dev))
        hsig = (ps_norm / denom < (1.4 + 2 / (d + 1)) * self.chiN).to(torch.float32)
        pc = (1 - self.cc) * state.pc + self._wide(hsig) * math.sqrt(self.cc * (2 - self.cc) * self.mueff) * delta / sigma_w
        sigma = state.sigma * torch.exp((self.cs / self.damps) * (ps_norm / self.chiN - 1))
        return ps, pc, self._wide(hsig), sigma

    def graph_signature(self, state):
        return ()

    def graph_host_counters(self):
        return ("count_iter",)                 # only counts; the device work reads count_iter_dev

    def _rows(self, state, sample_key, rows: int, row_offset: int):
        # noise rows addressed by their index in the whole population: a rank's rows are exactly those rows of the
        # single-process sample
        from ....ops import es as _es
        noise = _es.noise_rows(sample_key, rows, self.dim, row_offset)
        return state.mean + self._wide(state.sigma) * torch.sqrt(state.C) * noise

    def ask(self, state):
        key, sample_key = rnd.split(state.key)
        population = self._rows(state, sample_key, self.pop_size, 0)
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=state.count_iter_dev + 1, key=key)

    def _whiten(self, state, delta):
        return delta / torch.sqrt(state.C)

    def _update(self, state, step_sum, ysq_sum):
        """``step_sum = Σ w_i (x_i − m)``, ``ysq_sum = Σ w_i ((x_i − m)/σ)²`` over the selected individuals."""
        mean = state.mean + self.cm * step_sum
        ps, pc, hsig, sigma = self._paths(state, mean)
        C = ((1 - self.c1 - self.cmu) * state.C
             + self.c1 * (pc * pc + (1 - hsig) * self.cc * (2 - self.cc) * state.C)
             + self.cmu * ysq_sum)
        return state.replace(mean=mean, ps=ps, pc=pc, C=C, sigma=sigma)

    def tell(self, state, fitness):
        if self.block_fitness and fitness.dim() == 2:
            # every block selects its own best mu sub-vectors
            B, bd = self.blocks, self.block_dim
            order = torch.argsort(fitness, dim=0, stable=True)[: self.mu]                              # (mu, blocks)
            chosen = torch.gather(state.population.reshape(self.pop_size, B, bd), 0, order[:, :, None].expand(self.mu, B, bd))
            d = chosen.reshape(self.mu, self.dim) - state.mean
        else:
            order = _sort.argsort(fitness)
            d = state.population[order[: self.mu]] - state.mean
        y = d / self._wide(state.sigma)
        return self._update(state, self.weights @ d, self.weights @ (y * y))

    # Row-sharded: 2·dim floats per generation (the two weighted sums), nothing of size dim² anywhere.
    def ask_sharded(self, state, rank: int, world: int):
        if self.pop_size % world:
            raise ValueError(f"pop_size {self.pop_size} is not divisible by the number of ranks {world}")
        local = self.pop_size // world
        key, sample_key = rnd.split(state.key)
        population = self._rows(state, sample_key, local, rank * local)
        return population, state.replace(population=population, count_iter=state.count_iter + 1,
                                         count_iter_dev=state.count_iter_dev + 1, key=key)

    def tell_sharded(self, state, fitness, rank: int, world: int):
        from ....core import distributed as dist

        local = self.pop_size // world
        _, ranking = _sort.argsort_with_rank(fitness)
        padded = torch.cat([self.weights, self.weights.new_zeros(self.pop_size - self.mu)])
        w_local = padded[ranking[rank * local:(rank + 1) * local]]          # 0 outside the best μ
        d = state.population - state.mean
        y = d / self._wide(state.sigma)
        packed = dist.all_reduce(torch.cat([w_local @ d, w_local @ (y * y)]))
        return self._update(state, packed[: self.dim], packed[self.dim:])


class IPOPCMAES(CMAES):
    """CMA-ES with restarts and increasing population (Auger & Hansen 2005).

    A restart is triggered after ``stagnation_threshold`` generations without improvement of
    the best fitness; the run is re-initialised around the initial mean with the population
    multiplied by ``2**restarts``.  Because the population size changes, the restart decision
    is taken on the host (one scalar read-back per generation) and the state tensors are
    re-allocated.  (The reference's version never changes the population: its restart
    bookkeeping lives in Python attributes, ``cma_es.py:256-354``.)"""

    def __init__(self, center_init, init_stdev, pop_size=None, recombination_weights=None, cm=1,
                 stagnation_threshold=50, **kwargs):
        super().__init__(center_init, init_stdev, pop_size, recombination_weights, cm, **kwargs)
        self._ctor = dict(center_init=center_init, init_stdev=init_stdev, recombination_weights=None, cm=cm, **kwargs)
        self.original_pop_size = self.pop_size
        self.stagnation_threshold = stagnation_threshold

    ask_sharded = None
    tell_sharded = None

    def setup(self, key):
        s = super().setup(key)
        dev = self.center_init.device
        # best fitness and the stagnation counter live on the device: a generation without a restart then has no host
        # value that changes, and can be recorded; the restart decision reads the counter (every generation when
        # issued eagerly, at the workflow's polls — ``graph_retune`` — while replaying)
        return s.replace(best_fitness=torch.full((), float("inf"), device=dev),
                         stagnation_count=torch.zeros((), dtype=torch.int32, device=dev), restarts=0,
                         pop_size=self.original_pop_size)

    def _variant(self, pop_size) -> CMAES:
        # one instance per population size, kept: building one copies its weights to the device (a host-to-device copy
        # per generation otherwise, and nothing a recording can hold)
        cache = self.__dict__.setdefault("_variant_cache", {})
        if pop_size not in cache:
            cache[pop_size] = CMAES(pop_size=pop_size, **self._ctor)
        return cache[pop_size]

    def ask(self, state):
        if state.pop_size != self.pop_size:
            return self._variant(state.pop_size).ask(state)
        return super().ask(state)

    def _next_pop_size(self, state, restarts):
        return self.original_pop_size * (2 ** restarts)

    def tell(self, state, fitness):
        algo = self if state.pop_size == self.pop_size else self._variant(state.pop_size)
        state = CMAES.tell(algo, state, fitness)
        current = fitness.min()
        improved = current < state.best_fitness
        state = state.replace(best_fitness=torch.minimum(current, state.best_fitness),
                              stagnation_count=torch.where(improved, torch.zeros_like(state.stagnation_count),
                                                           state.stagnation_count + 1))
        if state.stagnation_count.is_cuda and torch.cuda.is_current_stream_capturing():
            return state                                   # recorded generation: the workflow polls ``graph_retune``
        return self._restart_if_stagnant(state)

    def graph_retune(self, state):
        """Polled by the workflow while it replays recorded generations: the restart that eager issue decides every
        generation (one scalar read-back) is decided here; a restart changes the population size, so the workflow
        records afresh."""
        fresh = self._restart_if_stagnant(state)
        return None if fresh is state else fresh

    def graph_variant(self, state):
        return state.pop_size

    def _restart_if_stagnant(self, state):
        if int(state.stagnation_count) < self.stagnation_threshold:
            return state
        restarts = state.restarts + 1
        new_pop = self._next_pop_size(state, restarts)
        key, sub = rnd.split(state.key)
        fresh = self._variant(new_pop).setup(sub)
        fields = {k: getattr(fresh, k) for k in fresh.keys()}
        fields.update(key=key, best_fitness=state.best_fitness, restarts=restarts,
                      stagnation_count=torch.zeros_like(state.stagnation_count), pop_size=new_pop)
        return state.replace(**fields)


class BIPOPCMAES(IPOPCMAES):
    """Restart schedule alternating large and small populations: the population doubles per
    restart until it exceeds 16× the original, then falls back to the original size
    (regime rule of the reference, ``cma_es.py:357-388``)."""

    def _next_pop_size(self, state, restarts):
        if state.pop_size > 16 * self.original_pop_size:
            return self.original_pop_size
        return state.pop_size * 2
