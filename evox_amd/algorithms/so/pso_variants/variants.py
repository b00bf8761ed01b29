"""PSO variants: CLPSO, SL-PSO (Gaussian / uniform demonstrator sampling), DMS-PSO-EL, FIPS,
FS-PSO and small-world PSO.  Reference behaviour: ``pso_variants/{clpso,sl_pso_gs,sl_pso_us,
dms_pso_el,fips,fs_pso,swmmpso}.py``.  All of them hand out the current swarm in ``ask`` and do
their whole update in ``tell``."""
from __future__ import annotations

import math

import torch

from ....core import random as rnd
from ....core.algorithm import Algorithm
from ....core.state import State
from ....ops import swarm as _swarm
from . import topology_utils as topo
from .utils import get_distance_matrix, init_swarm, update_bests


class _SwarmBase(Algorithm):
    def _bounds(self, lb, ub, pop_size):
        super().__init__()
        self.lb = torch.as_tensor(lb, dtype=torch.float32)
        self.ub = torch.as_tensor(ub, dtype=torch.float32).to(self.lb.device)
        self.dim = self.lb.shape[0]
        self.pop_size = int(pop_size)

    def ask(self, state):
        return state.population, state

    def _inf(self, n=None):
        return torch.full((self.pop_size if n is None else n,), float("inf"), device=self.lb.device)

    def _fused(self, state, *coefficients):
        """The velocity / position move runs as one launch (``ops.swarm.attract``) on the GPU when every coefficient is a
        host number; the tensor formulation next to each call is the host path and the oracle of the kernel's tests."""
        return _swarm.attract_supported(state.population) and all(isinstance(c, (int, float)) for c in coefficients)


class CLPSO(_SwarmBase):
    """Comprehensive-learning PSO (Liang et al. 2006): with probability ``P_c[i]`` particle i
    learns from the better of two random particles' personal bests instead of its own."""

    def __init__(self, lb, ub, pop_size, inertia_weight, const_coefficient, learning_probability):
        self._bounds(lb, ub, pop_size)
        self.w, self.c = inertia_weight, const_coefficient
        self.P_c = torch.as_tensor(learning_probability, dtype=torch.float32).to(self.lb.device)

    def setup(self, key):
        sk, kp, kv = rnd.split(key.to(self.lb.device), 3)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size)
        return State(population=pop, velocity=vel, pbest_position=pop, pbest_fitness=self._inf(),
                     gbest_position=pop[0], gbest_fitness=self._inf(1), key=sk)

    def tell(self, state, fitness):
        key, k_c, k_1, k_2, k_p = rnd.split(state.key, 5)
        n = self.pop_size
        pbest, pbest_fit, gbest, gbest_fit = update_bests(state.population, fitness, state.pbest_position,
                                                          state.pbest_fitness, state.gbest_position,
                                                          state.gbest_fitness)
        r1, r2 = rnd.randint(k_1, (n,), 0, n), rnd.randint(k_2, (n,), 0, n)
        exemplar = torch.where(pbest_fit[r1] < pbest_fit[r2], r1, r2)
        learn = rnd.uniform(k_p, (n,)) < self.P_c
        # as in the reference, the exemplar position is read from the previous personal bests
        if self._fused(state, self.w, self.c):
            # one launch: the particle's attractor is a row of the previous personal bests, its own or the exemplar's
            rows = torch.where(learn, exemplar, torch.arange(n, device=exemplar.device))
            pop, vel = _swarm.attract(state.population, state.velocity, self.lb, self.ub,
                                      [(state.pbest_position, rows, self.c, k_c, 0)], inertia=self.w)
        else:
            target = torch.where(learn[:, None], state.pbest_position[exemplar], state.pbest_position)
            vel = self.w * state.velocity + self.c * rnd.uniform(k_c, (n, self.dim)) * (target - state.population)
            pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(population=pop, velocity=vel, pbest_position=pbest, pbest_fitness=pbest_fit,
                             gbest_position=gbest, gbest_fitness=gbest_fit, key=key)


class _SLPSO(_SwarmBase):
    """Social-learning PSO (Cheng & Jin 2015): every particle imitates a *demonstrator* drawn
    from the swarm sorted by descending fitness, plus the swarm mean."""

    def __init__(self, lb, ub, pop_size, social_influence_factor, demonstrator_choice_factor):
        self._bounds(lb, ub, pop_size)
        self.social_influence_factor = social_influence_factor
        self.demonstrator_choice_factor = demonstrator_choice_factor

    def setup(self, key):
        sk, kp, kv = rnd.split(key.to(self.lb.device), 3)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size)
        return State(population=pop, velocity=vel, global_best_location=pop[0], global_best_fitness=self._inf(1),
                     key=sk)

    def _demonstrator_index(self, key):
        raise NotImplementedError

    def tell(self, state, fitness):
        key, k_r, k_d = rnd.split(state.key, 3)
        n = self.pop_size
        best_fit, i = torch.min(fitness, dim=0)               # 0-d tensors; no index is read back to the host
        take = best_fit < state.global_best_fitness.reshape(())
        gbest = torch.where(take, state.population.index_select(0, i.reshape(1))[0], state.global_best_location)
        gfit = torch.where(take, best_fit, state.global_best_fitness.reshape(())).reshape(1)
        order = torch.argsort(fitness, descending=True, stable=True)
        x_avg = state.population.mean(0)
        if self._fused(state, self.social_influence_factor):
            nd = n * self.dim                                   # the three factors are consecutive (n, dim) draws of k_r
            pop, vel = _swarm.attract(state.population, state.velocity, self.lb, self.ub,
                                      [(state.population, order[self._demonstrator_index(k_d)], 1.0, k_r, nd),
                                       (x_avg, "all", self.social_influence_factor, k_r, 2 * nd)],
                                      inertia=1.0, inertia_key=k_r, inertia_offset=0)
            return state.replace(population=pop, velocity=vel, global_best_location=gbest, global_best_fitness=gfit,
                                 key=key)
        r = rnd.uniform(k_r, (3, n, self.dim))
        x_k = state.population[order][self._demonstrator_index(k_d)]
        vel = r[0] * state.velocity + r[1] * (x_k - state.population) \
            + r[2] * self.social_influence_factor * (x_avg - state.population)
        pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(population=pop, velocity=vel, global_best_location=gbest, global_best_fitness=gfit,
                             key=key)


class SLPSOGS(_SLPSO):
    """Demonstrator rank drawn from a half-Gaussian anchored at the best particle."""

    def _demonstrator_index(self, key):
        n = self.pop_size
        ar = torch.arange(n, device=self.lb.device, dtype=torch.float32)
        sigma = self.demonstrator_choice_factor * (n - (ar + 1))
        z = rnd.normal(key, (n,))
        pos = sigma * (-z.abs()) + n
        return torch.floor(torch.clamp(pos, 1, n)).to(torch.int64) - 1


class SLPSOUS(_SLPSO):
    """Demonstrator rank drawn uniformly from the particles better than a rank-dependent bound."""

    def _demonstrator_index(self, key):
        n = self.pop_size
        ar = torch.arange(n, device=self.lb.device, dtype=torch.float32)
        q = torch.clamp(n - torch.ceil(self.demonstrator_choice_factor * (n - (ar + 1) - 1)), 1, n)
        u = rnd.uniform(key, (n,)) * (n + 1 - q) + q
        return torch.clamp(torch.floor(u).to(torch.int64) - 1, 0, n - 1)


class DMSPSOEL(_SwarmBase):
    """Dynamic multi-swarm PSO with elite learning: ``dynamic_sub_swarms_num`` small swarms that
    follow their local bests and are regrouped every ``regrouped_iteration_num`` iterations, a
    following swarm that learns from the regroup-time elite, and a final phase (last 10 % of
    ``max_iteration``) of plain global-best PSO."""

    def __init__(self, lb, ub, dynamic_sub_swarm_size, dynamic_sub_swarms_num, following_sub_swarm_size,
                 regrouped_iteration_num, max_iteration, inertia_weight, pbest_coefficient, lbest_coefficient,
                 rbest_coefficient, gbest_coefficient):
        self._bounds(lb, ub, dynamic_sub_swarm_size * dynamic_sub_swarms_num + following_sub_swarm_size)
        self.dynamic_sub_swarm_size = dynamic_sub_swarm_size
        self.dynamic_sub_swarms_num = dynamic_sub_swarms_num
        self.following_sub_swarm_size = following_sub_swarm_size
        self.regrouped_iteration_num = regrouped_iteration_num
        self.max_iteration = max_iteration
        self.w = inertia_weight
        self.c_pbest, self.c_lbest = pbest_coefficient, lbest_coefficient
        self.c_rbest, self.c_gbest = rbest_coefficient, gbest_coefficient

    def setup(self, key):
        dev = self.lb.device
        sk, kp, kv = rnd.split(key.to(dev), 3)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size)
        nd = self.dynamic_sub_swarm_size * self.dynamic_sub_swarms_num
        lbest = pop[:nd].reshape(self.dynamic_sub_swarms_num, self.dynamic_sub_swarm_size, self.dim)[:, 0, :]
        return State(iteration=0, population=pop, velocity=vel, pbest_position=pop, pbest_fitness=self._inf(),
                     lbest_position=lbest, lbest_fitness=self._inf(self.dynamic_sub_swarms_num),
                     rbest_index=torch.zeros(self.following_sub_swarm_size, dtype=torch.int64, device=dev),
                     gbest_position=torch.zeros(self.dim, device=dev),
                     gbest_fitness=torch.tensor(float("inf"), device=dev), key=sk)

    def graph_host_counters(self):
        return ("iteration",)

    def graph_variant(self, state):
        # the branches of tell(): which phase, and whether this iteration regroups the swarms
        multi = state.iteration < 0.9 * self.max_iteration
        return (multi, multi and state.iteration % self.regrouped_iteration_num == 0)

    def tell(self, state, fitness):
        if state.iteration < 0.9 * self.max_iteration:
            return self._multi_swarm_phase(state, fitness)
        return self._global_phase(state, fitness)

    def _regroup(self, state, fitness):
        nd = self.dynamic_sub_swarm_size * self.dynamic_sub_swarms_num
        order = torch.argsort(fitness, stable=True)
        key, k_shuffle = rnd.split(state.key)
        head = rnd.permutation(k_shuffle, order[:nd])
        idx = torch.cat([head, order[nd:]])
        rbest = torch.argsort(fitness[:nd], stable=True)[: self.following_sub_swarm_size]
        return state.replace(population=state.population[idx], velocity=state.velocity[idx],
                             pbest_position=state.pbest_position[idx], pbest_fitness=state.pbest_fitness[idx],
                             rbest_index=rbest, key=key)

    def _multi_swarm_phase(self, state, fitness):
        if state.iteration % self.regrouped_iteration_num == 0:
            state = self._regroup(state, fitness)
        key, k_p, k_l, k_r = rnd.split(state.key, 4)
        ns, ss, nd = self.dynamic_sub_swarms_num, self.dynamic_sub_swarm_size, self.dynamic_sub_swarm_size * self.dynamic_sub_swarms_num
        better = fitness < state.pbest_fitness
        pbest = torch.where(better[:, None], state.population, state.pbest_position)
        pbest_fit = torch.minimum(state.pbest_fitness, fitness)
        dyn_pos = state.population[:nd].reshape(ns, ss, self.dim)
        dyn_fit = fitness[:nd].reshape(ns, ss)
        lbest_fit, lbest_i = dyn_fit.min(1)
        lbest = dyn_pos[torch.arange(ns, device=dyn_pos.device), lbest_i]
        if self._fused(state, self.w, self.c_pbest, self.c_lbest, self.c_rbest):
            # two launches: the dynamic swarms follow their own and their swarm's best, the followers their own and a
            # regroup-time elite; both write their rows of the new swarm
            x, v = state.population, state.velocity
            pop, vel = torch.empty_like(x), torch.empty_like(v)
            swarm_of = torch.arange(nd, device=x.device) // ss
            _swarm.attract(x[:nd], v[:nd], self.lb, self.ub,
                           [(pbest[:nd], None, self.c_pbest, k_p, 0), (lbest, swarm_of, self.c_lbest, k_l, 0)],
                           inertia=self.w, out=(pop[:nd], vel[:nd]))
            if self.following_sub_swarm_size > 0:
                _swarm.attract(x[nd:], v[nd:], self.lb, self.ub,
                               [(pbest[nd:], None, self.c_pbest, k_p, nd * self.dim), (x, state.rbest_index, self.c_rbest, k_r, 0)],
                               inertia=self.w, out=(pop[nd:], vel[nd:]))
            return state.replace(iteration=state.iteration + 1, population=pop, velocity=vel, pbest_position=pbest,
                                 pbest_fitness=pbest_fit, lbest_position=lbest, lbest_fitness=lbest_fit, key=key)
        rbest = state.population[state.rbest_index]
        rp = rnd.uniform(k_p, (self.pop_size, self.dim))
        dyn_vel = self.w * state.velocity[:nd].reshape(ns, ss, self.dim) \
            + self.c_pbest * rp[:nd].reshape(ns, ss, self.dim) * (pbest[:nd].reshape(ns, ss, self.dim) - dyn_pos) \
            + self.c_lbest * rnd.uniform(k_l, (ns, ss, self.dim)) * (lbest[:, None, :] - dyn_pos)
        fol_pos = state.population[nd:]
        fol_vel = self.w * state.velocity[nd:] + self.c_pbest * rp[nd:] * (pbest[nd:] - fol_pos) \
            + self.c_rbest * rnd.uniform(k_r, (self.following_sub_swarm_size, self.dim)) * (rbest - fol_pos)
        vel = torch.cat([dyn_vel.reshape(nd, self.dim), fol_vel], dim=0)
        pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(iteration=state.iteration + 1, population=pop, velocity=vel, pbest_position=pbest,
                             pbest_fitness=pbest_fit, lbest_position=lbest, lbest_fitness=lbest_fit, key=key)

    def _global_phase(self, state, fitness):
        key, k_p, k_g = rnd.split(state.key, 3)
        better = fitness < state.pbest_fitness
        pbest = torch.where(better[:, None], state.population, state.pbest_position)
        pbest_fit = torch.minimum(state.pbest_fitness, fitness)
        g = torch.argmin(pbest_fit).reshape(1)              # gathered on the device: no index read-back
        gbest, gfit = pbest.index_select(0, g)[0], pbest_fit.index_select(0, g)[0]
        shape = (self.pop_size, self.dim)
        if self._fused(state, self.w, self.c_pbest, self.c_gbest):
            pop, vel = _swarm.attract(state.population, state.velocity, self.lb, self.ub,
                                      [(pbest, None, self.c_pbest, k_p, 0), (gbest, "all", self.c_gbest, k_g, 0)],
                                      inertia=self.w)
        else:
            vel = self.w * state.velocity + self.c_pbest * rnd.uniform(k_p, shape) * (pbest - state.population) \
                + self.c_gbest * rnd.uniform(k_g, shape) * (gbest - state.population)
            pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(iteration=state.iteration + 1, population=pop, velocity=vel, pbest_position=pbest,
                             pbest_fitness=pbest_fit, gbest_position=gbest, gbest_fitness=gfit, key=key)


def _constriction(phi):
    return 2 / (phi - 2 + torch.sqrt(torch.abs(phi * (phi - 4))))


class FIPS(_SwarmBase):
    """Fully informed particle swarm (Mendes, Kennedy & Neves 2004): a particle is attracted to
    the randomly weighted centroid of *all* its neighbours' personal bests.  Neighbour weights:
    "Constant", "Pbest" (1/fitness) or "Distance".  The per-generation random factors have
    shape (N, K, dim) with K the maximum degree of the topology (the reference draws
    N × N × dim, ``fips.py:155-159``)."""

    def __init__(self, lb, ub, pop_size, max_phi=4.1, mean=None, stdev=None, topology="Square",
                 weight_type="Distance", shortcut=0):
        self._bounds(lb, ub, pop_size)
        self.max_phi, self.mean, self.stdev = max_phi, mean, stdev
        self.topology, self.weight_type, self.shortcut = topology, weight_type, shortcut

    def setup(self, key):
        dev = self.lb.device
        sk, kp, kv = rnd.split(key.to(dev), 3)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size, self.mean, self.stdev)
        if self.topology in ("Square", "USquare"):
            adj = topo.get_square_neighbour(pop)
        elif self.topology in ("All", "UAll"):
            adj = topo.get_full_neighbour(pop)
        elif self.topology in ("Ring", "URing"):
            adj = topo.get_ring_neighbour(pop, 1)
        else:
            raise NotImplementedError(self.topology)
        phi = torch.full((self.pop_size, 1), float(self.max_phi), device=dev)
        # the topology is fixed after setup: its padded neighbour lists are built once, not every generation
        nbr, nbr_mask = topo.build_adjacancy_list_from_matrix(adj)
        return State(population=pop, velocity=vel, local_best_location=pop, local_best_fitness=self._inf(),
                     neighbour_best_location=pop, neighbour_best_fitness=self._inf(), adjacancy_matrix=adj,
                     adjacancy_list=nbr, adjacancy_mask=nbr_mask, key=sk, chi=_constriction(phi), phi=phi)

    def tell(self, state, fitness):
        key, k_phi = rnd.split(state.key)
        better = fitness < state.local_best_fitness
        lbest = torch.where(better[:, None], state.population, state.local_best_location)
        lbest_fit = torch.minimum(state.local_best_fitness, fitness)
        nbr, mask = state.adjacancy_list, state.adjacancy_mask
        n, k = nbr.shape
        if self.weight_type == "Constant":
            w = torch.ones((n, k), device=nbr.device)
        elif self.weight_type == "Pbest":
            w = 1.0 / lbest_fit[nbr]
        elif self.weight_type == "Distance":
            w = torch.linalg.vector_norm(lbest[nbr] - lbest[:, None, :], dim=-1)
        else:
            raise ValueError(self.weight_type)
        if self._fused(state, self.max_phi):
            pop, vel = _swarm.fips_move(k_phi, state.population, state.velocity, lbest, nbr, mask, w, state.chi, state.phi,
                                        self.lb, self.ub, self.max_phi)
        else:
            phik = rnd.uniform(k_phi, (n, k, self.dim)) * self.max_phi * mask[:, :, None].to(torch.float32)
            wphi = w[:, :, None] * phik
            pm = (lbest[nbr] * wphi).sum(1) / wphi.sum(1)
            vel = state.chi * (state.velocity + state.phi * (pm - state.population))
            pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(population=pop, velocity=vel, local_best_location=lbest, local_best_fitness=lbest_fit,
                             key=key)


class SwmmPSO(_SwarmBase):
    """Constriction PSO on a small-world neighbourhood: K-nearest "circles" with optional
    rewired shortcut edges; each particle follows its own and its neighbourhood's best."""

    def __init__(self, lb, ub, pop_size, max_phi_1=2.05, max_phi_2=2.05, max_phi=4.1, mean=None, stdev=None,
                 topology="Circles", shortcut=0):
        self._bounds(lb, ub, pop_size)
        self.max_phi_1, self.max_phi_2, self.max_phi = max_phi_1, max_phi_2, max_phi
        self.mean, self.stdev, self.topology, self.shortcut = mean, stdev, topology, shortcut

    def setup(self, key):
        dev = self.lb.device
        sk, kp, kv, ka = rnd.split(key.to(dev), 4)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size, self.mean, self.stdev)
        phi_val = self.max_phi if self.max_phi > 0 else self.max_phi_1 + self.max_phi_2
        phi = torch.full((self.pop_size, 1), float(phi_val), device=dev)
        if self.topology != "Circles":
            raise NotImplementedError(self.topology)
        adj = topo.get_circles_neighbour(ka, pop, K=2, shortcut=self.shortcut)
        nbr, nbr_mask = topo.build_adjacancy_list_from_matrix(adj)      # fixed topology: built once
        return State(population=pop, velocity=vel, local_best_location=pop, local_best_fitness=self._inf(),
                     neighbour_best_location=pop, neighbour_best_fitness=self._inf(), adjacancy_matrix=adj,
                     adjacancy_list=nbr, adjacancy_mask=nbr_mask, key=sk, chi=_constriction(phi), phi=phi)

    def tell(self, state, fitness):
        key, k1, k2 = rnd.split(state.key, 3)
        shape = (self.pop_size, self.dim)
        better = fitness < state.local_best_fitness
        lbest = torch.where(better[:, None], state.population, state.local_best_location)
        lbest_fit = torch.minimum(state.local_best_fitness, fitness)
        nbr = state.adjacancy_list
        nbest_fit, nbest_i = topo.get_neighbour_best_fitness(lbest_fit, nbr)
        nbest = lbest[nbest_i]
        if self._fused(state, self.max_phi_1, self.max_phi_2):
            pop, vel = _swarm.attract(state.population, state.velocity, self.lb, self.ub,
                                      [(lbest, None, self.max_phi_1, k1, 0), (lbest, nbest_i.reshape(-1), self.max_phi_2, k2, 0)],
                                      inertia=1.0, row_scale=state.chi)
        else:
            phi1 = rnd.uniform(k1, shape) * self.max_phi_1
            phi2 = rnd.uniform(k2, shape) * self.max_phi_2
            vel = state.chi * (state.velocity + phi1 * (lbest - state.population) + phi2 * (nbest - state.population))
            pop = torch.clamp(state.population + vel, self.lb, self.ub)
        return state.replace(population=pop, velocity=vel, local_best_location=lbest, local_best_fitness=lbest_fit,
                             neighbour_best_location=nbest, neighbour_best_fitness=nbest_fit, key=key)


class FSPSO(_SwarmBase):
    """Feature-selection style PSO/GA hybrid: the better half of the swarm moves by the PSO
    rule, the other half is refilled by binary tournaments among the elite followed by uniform
    reset mutation.  (The reference draws both tournament sides from one key, which makes every
    tournament a self-match, ``fs_pso.py:113-122``; two independent draws are used here.)"""

    def __init__(self, lb, ub, pop_size, inertia_weight=0.6, cognitive_coefficient=2.5, social_coefficient=0.8,
                 mean=None, stdev=None, mutate_rate=0.01):
        self._bounds(lb, ub, pop_size)
        self.w, self.phi_p, self.phi_g = inertia_weight, cognitive_coefficient, social_coefficient
        self.mean, self.stdev, self.mutate_rate = mean, stdev, mutate_rate

    def setup(self, key):
        sk, kp, kv = rnd.split(key.to(self.lb.device), 3)
        pop, vel = init_swarm(kp, kv, self.lb, self.ub, self.pop_size, self.mean, self.stdev)
        return State(population=pop, velocity=vel, local_best_location=pop, local_best_fitness=self._inf(),
                     global_best_location=pop[0], global_best_fitness=self._inf(1), key=sk)

    def tell(self, state, fitness):
        key, k_r, k_t, k_mu, k_ma = rnd.split(state.key, 5)
        half = self.pop_size // 2
        rest = self.pop_size - half
        elite = torch.argsort(fitness, stable=True)[:half]
        e_pop, e_vel, e_fit = state.population[elite], state.velocity[elite], fitness[elite]
        e_lb, e_lbfit = state.local_best_location[elite], state.local_best_fitness[elite]
        lbest, lbest_fit, gbest, gfit = update_bests(e_pop, e_fit, e_lb, e_lbfit, state.global_best_location,
                                                     state.global_best_fitness)
        if self._fused(state, self.w, self.phi_p, self.phi_g):
            new_pop, new_vel = _swarm.attract(e_pop, e_vel, self.lb, self.ub,
                                              [(e_lb, None, self.phi_p, k_r, 0), (gbest, "all", self.phi_g, k_r, half * self.dim)],
                                              inertia=self.w)
        else:
            r = rnd.uniform(k_r, (2, half, self.dim))
            new_vel = self.w * e_vel + self.phi_p * r[0] * (e_lb - e_pop) + self.phi_g * r[1] * (gbest - e_pop)
            new_pop = torch.clamp(e_pop + new_vel, self.lb, self.ub)
        t = rnd.randint(k_t, (2, rest), 0, half)
        pool = torch.where(e_fit[t[0]] < e_fit[t[1]], t[0], t[1])
        base = e_pop[pool]
        offset = (rnd.uniform(k_mu, (rest, self.dim)) * 2 - 1) * (self.ub - self.lb)
        mutate = rnd.uniform(k_ma, (rest, self.dim)) < self.mutate_rate
        off_pop = torch.clamp(base + torch.where(mutate, offset, torch.zeros_like(offset)), self.lb, self.ub)
        return state.replace(
            population=torch.cat([new_pop, off_pop]), velocity=torch.cat([new_vel, e_vel[pool]]),
            local_best_location=torch.cat([lbest, off_pop]), local_best_fitness=torch.cat([lbest_fit, self._inf(rest)]),
            global_best_location=gbest, global_best_fitness=gfit, key=key)
