"""Swarm updates.  ``pso_step`` is the whole PSO ``tell`` (personal/global best update, velocity
and position move with two uniform random factors per coordinate drawn from the Philox stream
of ``key``) — on a GPU it is two kernel launches instead of ~25 elementwise ops."""
from __future__ import annotations

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of

_native.register_hip_api({
    "evox_hip_cso_step": (_native.I, [_native.P] * 6 + [_native.F, _native.P, _native.P, _native.I, _native.I] + [_native.P] * 5),
    "evox_hip_pso_step": (_native.I, [_native.P] * 17 + [_native.I64, _native.I, _native.F, _native.F, _native.F,
                                                        _native.P]),
})


def pso_step_reference(key, pop, vel, fit, pbest, pbest_fit, gbest, gbest_fit, lb, ub, w, phi_p, phi_g):
    n, d = pop.shape
    r = rnd.uniform(key, (2, n, d))
    better = fit < pbest_fit
    pbest = torch.where(better[:, None], pop, pbest)
    pbest_fit = torch.where(better, fit, pbest_fit)
    i = torch.argmin(pbest_fit)
    take = pbest_fit[i] < gbest_fit
    gbest = torch.where(take, pbest[i], gbest)
    gbest_fit = torch.where(take, pbest_fit[i], gbest_fit).reshape(gbest_fit.shape)
    vel = w * vel + phi_p * r[0] * (pbest - pop) + phi_g * r[1] * (gbest - pop)
    pop = torch.clamp(pop + vel, lb, ub)
    return pop, vel, pbest, pbest_fit, gbest, gbest_fit


def pso_step(key, pop, vel, fit, pbest, pbest_fit, gbest, gbest_fit, lb, ub, w, phi_p, phi_g):
    if not on_gpu(pop):
        return pso_step_reference(key, pop, vel, fit, pbest, pbest_fit, gbest, gbest_fit, lb, ub, w, phi_p, phi_g)
    n, d = pop.shape
    pop, vel, pbest = pop.contiguous(), vel.contiguous(), pbest.contiguous()
    fit = fit.contiguous().float()
    o_pop, o_vel, o_pbest = torch.empty_like(pop), torch.empty_like(vel), torch.empty_like(pbest)
    o_pfit = torch.empty_like(pbest_fit)
    o_gbest, o_gfit = torch.empty_like(gbest), torch.empty_like(gbest_fit)
    ws = torch.empty(2, dtype=torch.int32, device=pop.device)
    check(hip().evox_hip_pso_step(
        key.data_ptr(), pop.data_ptr(), vel.data_ptr(), fit.data_ptr(), pbest.data_ptr(),
        pbest_fit.contiguous().data_ptr(), gbest.contiguous().data_ptr(), gbest_fit.data_ptr(),
        lb.contiguous().data_ptr(), ub.contiguous().data_ptr(),
        o_pop.data_ptr(), o_vel.data_ptr(), o_pbest.data_ptr(), o_pfit.data_ptr(), o_gbest.data_ptr(),
        o_gfit.data_ptr(), ws.data_ptr(), n, d, float(w), float(phi_p), float(phi_g), stream_of(pop)), "pso_step")
    return o_pop, o_vel, o_pbest, o_pfit, o_gbest, o_gfit


def cso_step(key, pop, vel, fit, perm, phi, lb, ub):
    """One competitive-swarm generation on the GPU (``csrc/swarm.hip``): ``perm`` pairs the first
    half of a random permutation with the second; returns ``(cand, population, velocity, students)``
    like the tensor formulation in ``algorithms/so/pso_variants/cso.py``."""
    n, d = pop.shape
    half = n // 2
    pop, vel = pop.contiguous(), vel.contiguous()
    new_pop, new_vel = pop.clone(), vel.clone()
    cand = torch.empty((half, d), dtype=torch.float32, device=pop.device)
    students = torch.empty(half, dtype=torch.int64, device=pop.device)
    center = pop.mean(0) if phi != 0 else None
    check(hip().evox_hip_cso_step(key.contiguous().data_ptr(), pop.data_ptr(), vel.data_ptr(), fit.contiguous().data_ptr(),
                                  perm.contiguous().data_ptr(), center.data_ptr() if center is not None else None,
                                  float(phi), lb.contiguous().data_ptr(), ub.contiguous().data_ptr(), half, d,
                                  new_pop.data_ptr(), new_vel.data_ptr(), cand.data_ptr(), students.data_ptr(),
                                  stream_of(pop)), "cso_step")
    return cand, new_pop, new_vel, students
