"""Swarm updates.  ``pso_step`` is the whole PSO ``tell`` (personal/global best update, velocity
and position move with two uniform random factors per coordinate drawn from the Philox stream
of ``key``) — on a GPU it is two kernel launches instead of ~25 elementwise ops."""
from __future__ import annotations

import torch

from .. import _native
from ..core import random as rnd
from ._util import check, hip, on_gpu, stream_of

_native.register_hip_api({
    "evox_hip_cso_step": (_native.I, [_native.P] * 6 + [_native.F, _native.P, _native.P, _native.I, _native.I, _native.I]
                          + [_native.P] * 5),
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
    ws = torch.empty(2 + 2048, dtype=torch.int32, device=pop.device)   # arg-min, improved flag, the spread scan's partial minima
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
    # an even swarm: every particle is in a pair and the kernel writes winners' rows too; otherwise the particle left
    # out keeps its row through the copies
    write_all = 2 * half == n and perm.numel() == n
    new_pop, new_vel = (torch.empty_like(pop), torch.empty_like(vel)) if write_all else (pop.clone(), vel.clone())
    cand = torch.empty((half, d), dtype=torch.float32, device=pop.device)
    students = torch.empty(half, dtype=torch.int64, device=pop.device)
    center = pop.mean(0) if phi != 0 else None
    check(hip().evox_hip_cso_step(key.contiguous().data_ptr(), pop.data_ptr(), vel.data_ptr(), fit.contiguous().data_ptr(),
                                  perm.contiguous().data_ptr(), center.data_ptr() if center is not None else None,
                                  float(phi), lb.contiguous().data_ptr(), ub.contiguous().data_ptr(), half, d, int(write_all),
                                  new_pop.data_ptr(), new_vel.data_ptr(), cand.data_ptr(), students.data_ptr(),
                                  stream_of(pop)), "cso_step")
    return cand, new_pop, new_vel, students


_native.register_hip_api({
    "evox_hip_swarm_attract": (_native.I, [_native.P] * 7 + [_native.I64, _native.I, _native.F, _native.P, _native.I64, _native.I]
                               + [_native.P] * 6 + [_native.P]),
})


def attract_supported(x: torch.Tensor) -> bool:
    """The fused move is taken on the GPU for fp32 swarms (``EVOX_AMD_SWARM_ATTRACT=0`` keeps the tensor formulation)."""
    import os

    return on_gpu(x) and x.dtype == torch.float32 and os.environ.get("EVOX_AMD_SWARM_ATTRACT", "1") != "0"


def attract(x, v, lb, ub, terms, inertia=1.0, inertia_key=None, inertia_offset=0, row_scale=None, out=None):
    """The move shared by the swarm variants in one launch (``csrc/swarm.hip::swarm_attract_kernel``)::

        v' = s_i · ( inertia · u0 · v + Σ_k (coef_k · u_k) · (table_k[row_k(i)] − x) ),    x' = clip(x + v', lb, ub)

    ``terms``: up to three ``(table, rows, coef, key, offset)``: ``rows`` is None (particle i reads row i), an int64 index
    per particle, or ``"all"`` (``table`` is one row); ``u_k`` is element ``offset + i·dim + c`` of the uniform stream of
    ``key`` (what ``random.uniform(key, (…, n, dim))`` puts there), or 1 when ``key`` is None.  ``inertia_key`` gives the
    inertia a uniform factor of its own, ``row_scale`` (n values) multiplies the whole new velocity.
    ``out``: optional ``(x', v')`` buffers (e.g. row slices of a larger swarm).  Returns ``(x', v')``."""
    import ctypes

    n, d = x.shape
    keep = [x.contiguous(), v.contiguous(), lb.contiguous(), ub.contiguous()]
    ox, ov = out if out is not None else (torch.empty((n, d), dtype=torch.float32, device=x.device),
                                          torch.empty((n, d), dtype=torch.float32, device=x.device))
    if not (ox.is_contiguous() and ov.is_contiguous()):
        raise ValueError("the output buffers must be contiguous")
    k = len(terms)
    tables, indices, keys = (ctypes.c_void_p * max(k, 1))(), (ctypes.c_void_p * max(k, 1))(), (ctypes.c_void_p * max(k, 1))()
    offsets, modes, coefs = (ctypes.c_int64 * max(k, 1))(), (ctypes.c_int * max(k, 1))(), (ctypes.c_float * max(k, 1))()
    for j, (table, rows, coef, key, offset) in enumerate(terms):
        t = table.contiguous()
        keep.append(t)
        tables[j] = t.data_ptr()
        if rows is None:
            if t.shape != (n, d):
                raise ValueError("a table read row by row must have the swarm's shape")
            modes[j] = 0
        elif isinstance(rows, str):
            if t.numel() != d:
                raise ValueError("a table shared by all particles must be one row")
            modes[j] = 2
        else:
            r = rows.contiguous()
            if r.dtype != torch.int64 or r.shape != (n,) or t.dim() != 2 or t.shape[1] != d:
                raise ValueError("rows must be one int64 index per particle into a table of `dim` columns")
            keep.append(r)
            indices[j] = r.data_ptr()
            modes[j] = 1
        if key is not None:
            kk = key.contiguous()
            keep.append(kk)
            keys[j] = kk.data_ptr()
        offsets[j], coefs[j] = int(offset), float(coef)
    ik = inertia_key.contiguous() if inertia_key is not None else None
    rs = row_scale.contiguous().reshape(-1) if row_scale is not None else None
    if rs is not None and rs.numel() != n:
        raise ValueError("row_scale must hold one value per particle")
    check(hip().evox_hip_swarm_attract(
        keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), rs.data_ptr() if rs is not None else None,
        ox.data_ptr(), ov.data_ptr(), n, d, float(inertia), ik.data_ptr() if ik is not None else None, int(inertia_offset), k,
        tables, indices, keys, offsets, modes, coefs, stream_of(x)), "swarm_attract")
    return ox, ov


_native.register_hip_api({
    "evox_hip_fips_move": (_native.I, [_native.P] * 13 + [_native.I64, _native.I, _native.I, _native.F, _native.P]),
})


def fips_move(key, x, v, pbest, nbr, mask, w, chi, phi, lb, ub, max_phi):
    """The fully informed move in one launch (``csrc/swarm.hip::fips_move_kernel``): ``nbr`` / ``mask`` / ``w`` are the
    padded ``(n, K)`` neighbour lists, their validity and the neighbour weights, ``chi`` / ``phi`` one value per particle;
    the random factors are the ``(n, K, dim)`` uniform draw of ``key``.  Returns ``(x', v')``."""
    n, d = x.shape
    K = nbr.shape[1]
    ox, ov = torch.empty((n, d), dtype=torch.float32, device=x.device), torch.empty((n, d), dtype=torch.float32, device=x.device)
    t = [a.contiguous() for a in (key, x, v, pbest, nbr, mask.to(torch.float32), w.to(torch.float32), chi.reshape(-1),
                                  phi.reshape(-1), lb, ub)]
    if t[4].dtype != torch.int64 or t[7].numel() != n or t[8].numel() != n or t[5].shape != (n, K) or t[6].shape != (n, K):
        raise ValueError("fips_move: neighbour lists (n, K) int64, one chi / phi per particle")
    check(hip().evox_hip_fips_move(*[a.data_ptr() for a in t], ox.data_ptr(), ov.data_ptr(), n, K, d, float(max_phi),
                                   stream_of(x)), "fips_move")
    return ox, ov
