"""Binding of the spatial-physics step kernel (``csrc/spatial.hip``): one environment per lane, all sub-steps of an
environment step in one launch.  The parameter block is the engine's model at fixed offsets."""
from __future__ import annotations

import os

import torch

from .. import _native
from ._util import check, hip, on_gpu, stream_of

P, I = _native.P, _native.I
_native.register_hip_api({
    "evox_hip_spatial_param_floats": (I, []),
    "evox_hip_spatial_step": (I, [P, I, I, I, P, P, P, P, P, I, P, P, P, P]),
})

MAX_BODIES, MAX_CONTACTS, MAX_ACTUATORS = 9, 12, 8
_OFF_BODY = 16
_OFF_ANC = _OFF_BODY + 16 * MAX_BODIES
_OFF_JOINT = _OFF_ANC + MAX_BODIES * MAX_BODIES
_OFF_ACT = _OFF_JOINT + 8 * (MAX_BODIES - 1)
_OFF_CON = _OFF_ACT + 2 * MAX_ACTUATORS
_OFF_TASK = _OFF_CON + 4 * MAX_CONTACTS
_FLOATS = _OFF_TASK + 16


def fits(engine) -> bool:
    m = engine.model
    return engine.nb <= MAX_BODIES and len(m.contacts) <= MAX_CONTACTS and len(m.actuators) <= MAX_ACTUATORS


def applies(engine, pos: torch.Tensor) -> bool:
    """The kernel serves fp32 states on a GPU (``EVOX_AMD_SPATIAL=tensor`` keeps the tensor formulation)."""
    return (on_gpu(pos) and pos.dtype == torch.float32 and fits(engine)
            and os.environ.get("EVOX_AMD_SPATIAL", "kernel") == "kernel")


def param_block(engine, task: dict = None) -> torch.Tensor:
    """The model as the float block the kernel reads (host tensor).  ``task``: the recipe applied in the kernel
    (``ctrl_range, forward_weight, ctrl_cost, healthy_reward, z_range, vel_clip, frame_dt`` — the ant's)."""
    m = engine.model
    assert int(hip().evox_hip_spatial_param_floats()) == _FLOATS
    blk = torch.zeros(_FLOATS, dtype=torch.float32)
    head = [engine.nb, len(m.contacts), len(m.actuators), m.substeps, m.dt, m.max_velocity, m.gravity, m.contact_stiffness,
            m.contact_damping, m.friction, m.limit_stiffness, m.limit_damping]
    blk[:len(head)] = torch.tensor(head, dtype=torch.float32)
    for i, b in enumerate(m.bodies):
        o = _OFF_BODY + 16 * i
        blk[o] = float(max(b.parent, 0))
        blk[o + 1] = b.mass
        blk[o + 2:o + 5] = engine.com[i]
        blk[o + 5:o + 8] = engine.inertia[i]
        blk[o + 8:o + 11] = engine.joint_pos[i]
        blk[o + 11:o + 14] = engine.axis[i]
        blk[_OFF_ANC + MAX_BODIES * i:_OFF_ANC + MAX_BODIES * i + engine.nb] = engine.anc[i]
    big = 3.0e38
    for j in range(engine.nj):
        o = _OFF_JOINT + 8 * j
        blk[o] = max(float(engine.lo[j]), -big)
        blk[o + 1] = min(float(engine.hi[j]), big)
        blk[o + 2], blk[o + 3], blk[o + 4] = float(engine.damping[j]), float(engine.stiffness[j]), float(engine.armature[j])
    for a, (body, gear) in enumerate(m.actuators):
        blk[_OFF_ACT + 2 * a] = float(body - 1)
        blk[_OFF_ACT + 2 * a + 1] = float(gear)
    for c, (body, point) in enumerate(m.contacts):
        blk[_OFF_CON + 4 * c] = float(body)
        blk[_OFF_CON + 4 * c + 1:_OFF_CON + 4 * c + 4] = torch.tensor(point, dtype=torch.float32)
    if task is not None:
        vals = [1.0, task["ctrl_range"], task["forward_weight"], task["ctrl_cost"], task["healthy_reward"], task["z_range"][0],
                task["z_range"][1], task["vel_clip"], task["frame_dt"]]
        blk[_OFF_TASK:_OFF_TASK + len(vals)] = torch.tensor(vals, dtype=torch.float32)
    return blk


def _call(engine, params, pos, quat, theta, vel, ctrl, obs, total, done):
    m = engine.model
    nact = len(m.actuators)
    B = pos.shape[0]
    assert pos.shape == (B, 3) and quat.shape == (B, 4) and theta.shape == (B, engine.nj) and vel.shape == (B, engine.ndof)
    for t in (pos, quat, theta, vel):
        assert t.is_contiguous() and t.dtype == torch.float32
    if nact:
        ctrl = (torch.zeros(B, nact, device=pos.device) if ctrl is None else ctrl.to(torch.float32).contiguous())
        assert ctrl.shape == (B, nact)
    check(hip().evox_hip_spatial_step(params.data_ptr(), engine.nb, len(m.contacts), nact, pos.data_ptr(), quat.data_ptr(),
                                      theta.data_ptr() if engine.nj else None, vel.data_ptr(),
                                      ctrl.data_ptr() if nact else None, B,
                                      obs.data_ptr() if obs is not None else None,
                                      total.data_ptr() if total is not None else None,
                                      done.data_ptr() if done is not None else None, stream_of(pos)), "spatial_step")


def _params(engine, device, task=None):
    key = (device, task is not None)
    if key not in engine._params:
        engine._params[key] = param_block(engine, task).to(device)
    return engine._params[key]


def step(engine, pos, quat, theta, vel, ctrl):
    """One environment step for every row (new tensors)."""
    pos, quat, theta, vel = (t.contiguous().clone() for t in (pos, quat, theta, vel))
    _call(engine, _params(engine, pos.device), pos, quat, theta, vel, ctrl, None, None, None)
    return pos, quat, theta, vel


def step_task(engine, task: dict, pos, quat, theta, vel, ctrl, obs_dim: int, total: torch.Tensor, done: torch.Tensor):
    """One environment step with the task's recipe applied in the kernel.  IN PLACE: the state advances, ``total`` gains
    the step's reward where ``done`` was not set, ``done`` (bool) is set where the episode ends.  Returns the
    observations (batch, obs_dim)."""
    B = pos.shape[0]
    assert total.shape == (B,) and total.dtype == torch.float32 and total.is_contiguous()
    assert done.shape == (B,) and done.dtype == torch.bool and done.is_contiguous()
    obs = torch.empty((B, obs_dim), dtype=torch.float32, device=pos.device)
    _call(engine, _params(engine, pos.device, task), pos, quat, theta, vel, ctrl, obs, total, done)
    return obs
