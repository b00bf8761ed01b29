"""Binding of the planar-physics step kernel (``csrc/planar.hip``): one environment per lane, all sub-steps of an
environment step in one launch.  The parameter block is the engine's static matrices at fixed offsets."""
from __future__ import annotations

import os

import torch

from .. import _native
from ._util import check, hip, on_gpu, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({
    "evox_hip_planar_param_floats": (I, []),
    "evox_hip_planar_step": (I, [P, I, I, I, P, P, P, I, P]),
    "evox_hip_planar_step_task": (I, [P, I, I, I, P, P, P, I, P, P, P, P, I, P]),
})

MAX_LINKS, MAX_CONTACTS, MAX_ACTUATORS = 7, 8, 8
_OFF = dict(head=0, free=16, anc=32, wcom=96, wcon=160, mass=224, length=232, rest=240, mconst=256, lo=352, hi=368,
            damping=384, stiffness=400, act_dof=416, act_gear=424, task=440)


def fits(engine) -> bool:
    m = engine.model
    return engine.L <= MAX_LINKS and len(m.contacts) <= MAX_CONTACTS and len(m.actuators) <= MAX_ACTUATORS


def applies(engine, q: torch.Tensor) -> bool:
    """The kernel serves fp32 states on a GPU (``EVOX_AMD_PLANAR=tensor`` keeps the tensor formulation)."""
    return (on_gpu(q) and q.dtype == torch.float32 and fits(engine)
            and os.environ.get("EVOX_AMD_PLANAR", "kernel") == "kernel")


def param_block(engine, task: dict = None) -> torch.Tensor:
    """The engine's constants as the float block the kernel reads (host tensor).  ``task``: the recipe that
    ``step_task`` applies in the kernel — ``recipe`` 1 (locomotion: ``forward_weight, ctrl_cost, healthy_reward,
    terminate, z_range, angle_range, obs_skip, vel_clip, frame_dt``), 2 (cart with one pole: ``max_angle``), 3 (cart with
    two poles: ``tip``, the weight row of the tip) or 4 (two-link reacher: ``tip``), each with ``ctrl_range``."""
    m, L, N = engine.model, engine.L, engine.ndof
    blk = torch.zeros(int(hip().evox_hip_planar_param_floats()) if torch.cuda.is_available() else 464, dtype=torch.float32)
    if task is not None:
        big = 3.0e38
        vals = [float(task["recipe"]), task["ctrl_range"]]
        if task["recipe"] == 1:
            zr = task["z_range"] or (-big, big)
            ar = task["angle_range"] or (-big, big)
            vals += [task["forward_weight"], task["ctrl_cost"], task["healthy_reward"], 1.0 if task["terminate"] else 0.0,
                     max(zr[0], -big), min(zr[1], big), max(ar[0], -big), min(ar[1], big), float(task["obs_skip"]),
                     task["vel_clip"], task["frame_dt"]]
        elif task["recipe"] == 2:
            vals += [task["max_angle"]]
        blk[_OFF["task"]:_OFF["task"] + len(vals)] = torch.tensor(vals, dtype=torch.float32)
        if task["recipe"] in (3, 4):
            blk[_OFF["task"] + 16:_OFF["task"] + 16 + L] = task["tip"].reshape(-1)[:L]
    head = [L, len(m.contacts), len(m.actuators), m.substeps, m.dt, m.max_velocity, m.gravity, m.contact_stiffness,
            m.contact_damping, m.friction, m.limit_stiffness, m.limit_damping, m.drag[0], m.drag[1], m.drag[2]]
    blk[:len(head)] = torch.tensor(head, dtype=torch.float32)
    free = torch.zeros(N)
    free[engine.free] = 1.0
    blk[_OFF["free"]:_OFF["free"] + N] = free

    def put(off, mat, stride):
        for r in range(mat.shape[0]):
            blk[off + r * stride: off + r * stride + mat.shape[1]] = mat[r]

    put(_OFF["anc"], engine.anc, MAX_LINKS)
    put(_OFF["wcom"], engine.W_com, MAX_LINKS)
    if engine.W_con.shape[0]:
        put(_OFF["wcon"], engine.W_con, MAX_LINKS)
    blk[_OFF["mass"]:_OFF["mass"] + L] = engine.mass
    blk[_OFF["length"]:_OFF["length"] + L] = engine.lengths
    blk[_OFF["rest"]:_OFF["rest"] + L] = engine.rest_abs
    put(_OFF["mconst"], engine.M_const, MAX_LINKS + 2)
    big = 3.0e38                                     # stands for "no limit" (keeps inf − inf out of the kernel)
    blk[_OFF["lo"]:_OFF["lo"] + N] = engine.lo.clamp_min(-big)
    blk[_OFF["hi"]:_OFF["hi"] + N] = engine.hi.clamp_max(big)
    blk[_OFF["damping"]:_OFF["damping"] + N] = engine.damping
    blk[_OFF["stiffness"]:_OFF["stiffness"] + N] = engine.stiffness
    for a, (dof, gear) in enumerate(m.actuators):
        blk[_OFF["act_dof"] + a] = float(dof)
        blk[_OFF["act_gear"] + a] = float(gear)
    return blk


def step(engine, params: torch.Tensor, q: torch.Tensor, qd: torch.Tensor, ctrl):
    """One environment step for every row of ``q`` / ``qd`` (new tensors)."""
    m = engine.model
    q, qd = q.contiguous().clone(), qd.contiguous().clone()
    nact = len(m.actuators)
    if nact:
        ctrl = (torch.zeros(q.shape[0], nact, device=q.device) if ctrl is None else ctrl.to(torch.float32).contiguous())
        assert ctrl.shape == (q.shape[0], nact)
    assert q.shape == qd.shape == (q.shape[0], engine.ndof)
    check(hip().evox_hip_planar_step(params.data_ptr(), engine.L, len(m.contacts), nact, q.data_ptr(), qd.data_ptr(),
                                     ctrl.data_ptr() if nact else None, q.shape[0], stream_of(q)), "planar_step")
    return q, qd


def step_task(engine, params: torch.Tensor, recipe: int, q: torch.Tensor, qd: torch.Tensor, ctrl: torch.Tensor, obs_dim: int,
              total: torch.Tensor, done: torch.Tensor, aux: torch.Tensor = None) -> torch.Tensor:
    """One environment step with the task's recipe applied in the kernel.  IN PLACE: ``q`` and ``qd`` advance,
    ``total`` gains the step's reward where ``done`` was not set, ``done`` (bool) is set where the episode ends.  Returns
    the observations (batch, obs_dim).  ``aux``: per-environment data of the recipe (the reacher's targets)."""
    m = engine.model
    nact = len(m.actuators)
    B = q.shape[0]
    assert q.is_contiguous() and qd.is_contiguous() and q.shape == qd.shape == (B, engine.ndof)
    assert total.shape == (B,) and total.dtype == torch.float32 and total.is_contiguous()
    assert done.shape == (B,) and done.dtype == torch.bool and done.is_contiguous()
    ctrl = ctrl.to(torch.float32).contiguous()
    assert ctrl.shape == (B, nact) and nact > 0
    if aux is not None:
        aux = aux.to(torch.float32).contiguous()
        assert aux.shape == (B, 2)
    obs = torch.empty((B, obs_dim), dtype=torch.float32, device=q.device)
    check(hip().evox_hip_planar_step_task(params.data_ptr(), engine.L, len(m.contacts), nact, q.data_ptr(), qd.data_ptr(),
                                          ctrl.data_ptr(), B, obs.data_ptr(), total.data_ptr(), done.data_ptr(),
                                          aux.data_ptr() if aux is not None else None, int(recipe), stream_of(q)),
          "planar_step_task")
    return obs
