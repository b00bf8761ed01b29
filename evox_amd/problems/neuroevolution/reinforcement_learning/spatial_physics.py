"""Batched spatial (3-D) articulated-body physics on the device, and the ``ant`` task on it.

Role parity: the spatial environments the reference reaches through Brax (``src/evox/problems/neuroevolution/
reinforcement_learning/brax.py:15-133``; its own visualisation test drives ``ant``, ``tests/visualize/test_brax.py:8``).
The planar tasks live in ``planar_physics.py``; this is the same construction with rotations in space: a free-floating
root body and a tree of bodies on hinge joints, in generalized coordinates

    q = (p ∈ R³, unit quaternion of the root, θ₁ … θ_J)        q̇ = (v, ω (world frame), θ̇₁ … θ̇_J)

For every body the world pose follows from its parent's (one loop over the tree, batched over environments); the
Jacobian of a point r on body i has the columns  ``I₃`` (root translation), ``−[r − p]ₓ`` (root rotation) and
``a_k × (r − j_k)`` for every hinge k between the root and i (world axis ``a_k``, world position ``j_k``); the angular
Jacobian has ``I₃`` and ``a_k``.  With these

    M = Σ_i m_i J_vᵀ J_v + J_ωᵀ (R_i I_i R_iᵀ) J_ω
    Q = Σ_i J_vᵀ (m_i g + f_i − m_i a⁰_i) + J_ωᵀ (−I^w_i α⁰_i − ω_i × I^w_i ω_i) + joint torques

where ``a⁰, α⁰`` are the accelerations at q̈ = 0 (propagated down the tree).  ``M q̈ = Q`` is solved for all environments at
once and integrated semi-implicitly (the quaternion by its derivative, re-normalised).  Ground contact is a penalty
force at chosen points (spring + damper fading in with the penetration, smoothed Coulomb friction in the tangent
plane); joint limits are penalty torques.  Tensor code without host synchronisation; on a GPU the step is one launch
of ``csrc/spatial.hip`` (one environment per lane), for which this file is the oracle."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ....core import random as rnd
from .planar_physics import CONTACT_RAMP, LIMIT_RAMP
from .tensor_env import TensorEnv, register_tensor_env


@dataclass
class Body:
    """A rigid body.  The body frame sits at its joint (the root's at its reference point) and is aligned with the
    parent's frame at zero joint angle.  ``joint_pos``: the joint in the PARENT's frame; ``axis``: the hinge axis (unit,
    parent frame = own frame); ``com``: centre of mass in the own frame; ``inertia``: principal moments about the centre
    of mass, along the own frame's axes."""
    name: str
    parent: int
    mass: float
    com: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    inertia: Tuple[float, float, float] = (1e-3, 1e-3, 1e-3)
    joint_pos: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    axis: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    range: Optional[Tuple[float, float]] = None
    damping: float = 0.0
    armature: float = 0.0
    stiffness: float = 0.0


@dataclass
class SpatialModel:
    bodies: List[Body]
    gravity: float = 9.81
    actuators: List[Tuple[int, float]] = field(default_factory=list)          # (body whose joint is driven, gear)
    contacts: List[Tuple[int, Tuple[float, float, float]]] = field(default_factory=list)   # (body, point in its frame)
    contact_stiffness: float = 2.0e4
    contact_damping: float = 3.0e2
    friction: float = 1.0
    limit_stiffness: float = 1.0e3
    limit_damping: float = 1.0e1
    dt: float = 0.0025
    substeps: int = 4
    max_velocity: float = 100.0


def _skew(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1),
                        torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def quat_to_matrix(q):
    w, x, y, z = q.unbind(-1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def _axis_angle_matrix(axis, angle):
    """Rotation by ``angle`` (B,) about the unit ``axis`` (3,): Rodrigues."""
    K = _skew(axis)                                                           # (3, 3)
    s, c = torch.sin(angle)[:, None, None], torch.cos(angle)[:, None, None]
    eye = torch.eye(3, dtype=angle.dtype, device=angle.device)
    return eye + s * K + (1 - c) * (K @ K)


class SpatialEngine:
    """Static tensors of a :class:`SpatialModel` and one integration step for a batch of states.

    State: ``pos`` (B, 3), ``quat`` (B, 4: w, x, y, z), ``theta`` (B, J), ``vel`` (B, 3 + 3 + J: linear, angular in the
    world frame, joint rates).  J = bodies − 1 (body i ≥ 1 hangs on joint i − 1)."""

    def __init__(self, model: SpatialModel):
        self.model = model
        nb = len(model.bodies)
        self.nb, self.nj, self.ndof = nb, nb - 1, 6 + nb - 1
        for i, b in enumerate(model.bodies):
            assert b.parent < i and (i == 0) == (b.parent < 0), "bodies are listed parents first, the root first"
        anc = torch.zeros(nb, nb)
        for i in range(nb):
            k = i
            while k > 0:
                anc[i, k] = 1.0
                k = model.bodies[k].parent
        self.anc = anc                                          # anc[i, k] = 1 when joint of body k (k >= 1) lies between root and i
        f = lambda name: torch.tensor([getattr(b, name) for b in model.bodies], dtype=torch.float32)
        self.mass, self.com, self.inertia = f("mass"), f("com"), f("inertia")
        self.joint_pos, self.axis = f("joint_pos"), torch.nn.functional.normalize(f("axis"), dim=1)
        lo = torch.full((self.nj,), -float("inf"))
        hi = torch.full((self.nj,), float("inf"))
        for i, b in enumerate(model.bodies[1:]):
            if b.range is not None:
                lo[i], hi[i] = b.range
        self.lo, self.hi = lo, hi
        self.damping = torch.tensor([b.damping for b in model.bodies[1:]])
        self.stiffness = torch.tensor([b.stiffness for b in model.bodies[1:]])
        self.armature = torch.tensor([b.armature for b in model.bodies[1:]])
        gear = torch.zeros(len(model.actuators), self.nj)
        for a, (body, g) in enumerate(model.actuators):
            gear[a, body - 1] = g
        self.gear = gear
        self.con_body = torch.tensor([c[0] for c in model.contacts], dtype=torch.int64)
        self.con_local = torch.tensor([c[1] for c in model.contacts], dtype=torch.float32).reshape(-1, 3)
        self._dev: Dict = {}
        self._resident: Dict = {}
        self._params: Dict = {}

    _NAMES = ("anc", "mass", "com", "inertia", "joint_pos", "axis", "lo", "hi", "damping", "stiffness", "armature", "gear",
              "con_body", "con_local")

    def on(self, device, dtype=torch.float32):
        key = (device, dtype)
        if key not in self._dev:
            self._dev[key] = {n: (getattr(self, n).to(device) if n == "con_body" else getattr(self, n).to(device, dtype))
                              for n in self._NAMES}
        return self._dev[key]

    # -- kinematics ------------------------------------------------------------------------------------------------
    def kinematics(self, pos, quat, theta):
        """World rotation ``R`` (B, nb, 3, 3), frame origin ``x`` (B, nb, 3) and hinge axis ``a`` (B, nb, 3; row 0 unused)
        of every body."""
        c = self.on(pos.device, pos.dtype)
        R = [quat_to_matrix(quat)]
        x = [pos]
        a = [torch.zeros_like(pos)]
        for i in range(1, self.nb):
            p = self.model.bodies[i].parent
            a.append(torch.einsum("bij,j->bi", R[p], c["axis"][i]))
            x.append(x[p] + torch.einsum("bij,j->bi", R[p], c["joint_pos"][i]))
            R.append(R[p] @ _axis_angle_matrix(c["axis"][i], theta[:, i - 1]))
        return torch.stack(R, 1), torch.stack(x, 1), torch.stack(a, 1)

    def _jacobians(self, pos, x, a, points, body_of):
        """Linear Jacobians (B, P, 3, ndof) of world ``points`` (B, P, 3) fixed to the bodies ``body_of`` (P,)."""
        c = self.on(pos.device, pos.dtype)
        B, P = points.shape[0], points.shape[1]
        J = pos.new_zeros(B, P, 3, self.ndof)
        J[:, :, 0, 0] = 1.0
        J[:, :, 1, 1] = 1.0
        J[:, :, 2, 2] = 1.0
        J[:, :, :, 3:6] = -_skew(points - pos[:, None, :])
        if self.nj:
            rel = points[:, :, None, :] - x[:, None, 1:, :]                                 # (B, P, J, 3)
            cols = torch.cross(a[:, None, 1:, :].expand_as(rel), rel, dim=-1)               # a_k × (r − j_k)
            mask = c["anc"][body_of][:, 1:]                                                  # (P, J)
            J[:, :, :, 6:] = (cols * mask[None, :, :, None]).transpose(2, 3)
        return J

    # -- dynamics --------------------------------------------------------------------------------------------------
    def acceleration(self, pos, quat, theta, vel, ctrl):
        """Generalized accelerations (B, ndof) for controls ``ctrl`` (B, A)."""
        m, c = self.model, self.on(pos.device, pos.dtype)
        B = pos.shape[0]
        R, x, a = self.kinematics(pos, quat, theta)
        v0, w0, thd = vel[:, :3], vel[:, 3:6], vel[:, 6:]
        # velocities and the accelerations at q̈ = 0, down the tree
        w = [w0]
        alpha0 = [torch.zeros_like(w0)]
        ao0 = [torch.zeros_like(w0)]
        for i in range(1, self.nb):
            p = m.bodies[i].parent
            rate = thd[:, i - 1:i] * a[:, i]
            w.append(w[p] + rate)
            alpha0.append(alpha0[p] + torch.cross(w[p], rate, dim=-1))
            r = x[:, i] - x[:, p]
            ao0.append(ao0[p] + torch.cross(alpha0[p], r, dim=-1) + torch.cross(w[p], torch.cross(w[p], r, dim=-1), dim=-1))
        w, alpha0, ao0 = torch.stack(w, 1), torch.stack(alpha0, 1), torch.stack(ao0, 1)
        d = torch.einsum("bnij,nj->bni", R, c["com"])                                       # COM offsets in the world
        com = x + d
        ac0 = ao0 + torch.cross(alpha0, d, dim=-1) + torch.cross(w, torch.cross(w, d, dim=-1), dim=-1)
        bodies = torch.arange(self.nb, device=pos.device)
        Jv = self._jacobians(pos, x, a, com, bodies)                                        # (B, nb, 3, ndof)
        Jw = pos.new_zeros(B, self.nb, 3, self.ndof)
        Jw[:, :, 0, 3] = 1.0
        Jw[:, :, 1, 4] = 1.0
        Jw[:, :, 2, 5] = 1.0
        if self.nj:
            Jw[:, :, :, 6:] = (a[:, None, 1:, :] * c["anc"][:, 1:][None, :, :, None]).transpose(2, 3)
        Iw = torch.einsum("bnij,nj,bnkj->bnik", R, c["inertia"], R)                         # R diag(I) Rᵀ
        M = torch.einsum("n,bncd,bnce->bde", c["mass"], Jv, Jv) + torch.einsum("bncd,bnck,bnke->bde", Jw, Iw, Jw)
        if self.nj:
            M = M + torch.diag_embed(torch.cat([pos.new_zeros(6), c["armature"]]))
        grav = pos.new_tensor([0.0, 0.0, -m.gravity])
        force = c["mass"][None, :, None] * (grav - ac0)
        Iw_w = torch.einsum("bnij,bnj->bni", Iw, w)
        torque = -torch.einsum("bnij,bnj->bni", Iw, alpha0) - torch.cross(w, Iw_w, dim=-1)
        Q = torch.einsum("bncd,bnc->bd", Jv, force) + torch.einsum("bncd,bnc->bd", Jw, torque)
        if self.nj:
            below, above = (c["lo"] - theta).clamp_min(0.0), (theta - c["hi"]).clamp_min(0.0)
            outside = ((below + above) * (1.0 / LIMIT_RAMP)).clamp_max(1.0)
            tau = (-c["damping"] * thd - c["stiffness"] * theta + m.limit_stiffness * (below - above)
                   - m.limit_damping * outside * thd)
            if ctrl is not None and c["gear"].shape[0]:
                tau = tau + ctrl @ c["gear"]
            Q = torch.cat([Q[:, :6], Q[:, 6:] + tau], dim=1)
        if c["con_body"].numel():
            cb = c["con_body"]
            pts = x[:, cb] + torch.einsum("bpij,pj->bpi", R[:, cb], c["con_local"])
            Jp = self._jacobians(pos, x, a, pts, cb)
            pv = torch.einsum("bpcd,bd->bpc", Jp, vel)
            depth = (-pts[..., 2]).clamp_min(0.0)
            touching = (depth * (1.0 / CONTACT_RAMP)).clamp_max(1.0)
            fn = (m.contact_stiffness * depth - m.contact_damping * pv[..., 2] * touching).clamp_min(0.0)
            speed = torch.sqrt(pv[..., 0] ** 2 + pv[..., 1] ** 2 + 1e-12)
            ft = -m.friction * fn * torch.tanh(speed / 0.05) / speed
            f = torch.stack([ft * pv[..., 0], ft * pv[..., 1], fn], dim=-1)
            Q = Q + torch.einsum("bpcd,bpc->bd", Jp, f)
        return torch.cholesky_solve(Q[..., None], torch.linalg.cholesky_ex(M).L)[..., 0]

    def step_tensor(self, pos, quat, theta, vel, ctrl):
        dt, vmax = self.model.dt, self.model.max_velocity
        for _ in range(self.model.substeps):
            vel = (vel + dt * self.acceleration(pos, quat, theta, vel, ctrl)).clamp(-vmax, vmax)
            pos = pos + dt * vel[:, :3]
            w = vel[:, 3:6]
            qw, qv = quat[:, :1], quat[:, 1:]
            dq = 0.5 * torch.cat([-(w * qv).sum(-1, keepdim=True), qw * w + torch.cross(w, qv, dim=-1)], dim=1)
            quat = torch.nn.functional.normalize(quat + dt * dq, dim=1)
            theta = theta + dt * vel[:, 6:]
        return pos, quat, theta, vel

    def step(self, pos, quat, theta, vel, ctrl):
        """``substeps`` semi-implicit Euler steps under constant controls (the kernel on a GPU, see ``ops/spatial.py``)."""
        from ....ops import spatial as _k

        if _k.applies(self, pos):
            return _k.step(self, pos, quat, theta, vel, ctrl)
        return self.step_tensor(pos, quat, theta, vel, ctrl)

    # -- diagnostics -----------------------------------------------------------------------------------------------
    def momenta_energy(self, pos, quat, theta, vel):
        """Linear momentum (B, 3), angular momentum about the origin (B, 3), kinetic + potential energy (B,)."""
        c = self.on(pos.device, pos.dtype)
        R, x, a = self.kinematics(pos, quat, theta)
        w = [vel[:, 3:6]]
        vo = [vel[:, :3]]
        for i in range(1, self.nb):
            p = self.model.bodies[i].parent
            w.append(w[p] + vel[:, 6 + i - 1:6 + i] * a[:, i])
            vo.append(vo[p] + torch.cross(w[p], x[:, i] - x[:, p], dim=-1))
        w, vo = torch.stack(w, 1), torch.stack(vo, 1)
        d = torch.einsum("bnij,nj->bni", R, c["com"])
        com, vc = x + d, vo + torch.cross(w, d, dim=-1)
        Iw = torch.einsum("bnij,nj,bnkj->bnik", R, c["inertia"], R)
        Lw = torch.einsum("bnij,bnj->bni", Iw, w)
        mass = c["mass"][None, :, None]
        lin = (mass * vc).sum(1)
        ang = (torch.cross(com, mass * vc, dim=-1) + Lw).sum(1)
        energy = (0.5 * (mass[..., 0] * (vc * vc).sum(-1)) + 0.5 * (w * Lw).sum(-1)
                  + self.model.gravity * mass[..., 0] * com[..., 2]).sum(1)
        energy = energy + 0.5 * (c["armature"] * vel[:, 6:] ** 2).sum(1)
        return lin, ang, energy

    def segments(self, pos, quat, theta, index: int = 0):
        """For drawing: per body the world segment from its frame origin to twice its centre-of-mass offset (bodies are
        modelled as rods from the joint through the centre), projected on the x–z plane: (nb, 2, 2)."""
        c = self.on(pos.device, pos.dtype)
        R, x, _ = self.kinematics(pos[index:index + 1], quat[index:index + 1], theta[index:index + 1])
        end = x + 2.0 * torch.einsum("bnij,nj->bni", R, c["com"])
        seg = torch.stack([x[0], end[0]], dim=1)                                            # (nb, 2, 3)
        return seg[..., [0, 2]].detach().cpu()


# ---------------------------------------------------------------------------------------------------------------------
# ant
# ---------------------------------------------------------------------------------------------------------------------
def _rod_inertia(mass: float, vec: Sequence[float], radius: float):
    """Principal moments of a thin rod from the origin to ``vec``, expressed along the frame axes (exact for rods along
    an axis; a rod in the x–y plane gets the rod's moment about the vertical axis and about its own perpendicular)."""
    length = math.sqrt(sum(v * v for v in vec))
    along = 0.5 * mass * radius * radius
    across = mass * length * length / 12.0 + 0.25 * mass * radius * radius
    ux, uy, uz = (abs(v) / length for v in vec)
    # moment about axis e: along·(u·e)² + across·(1 − (u·e)²)
    return tuple(along * u * u + across * (1 - u * u) for u in (ux, uy, uz))


class Ant(TensorEnv):
    """Four-legged ant (observation 27 = height, orientation quaternion, 8 joint angles, 6 + 8 velocities; action 8):
    forward velocity + 1 per healthy step − half the squared controls; ends when the torso leaves 0.2–1.0 m.  Torso
    sphere of radius 0.25 with four legs of two hinged segments each (hip about the vertical, ankle about the horizontal
    axis across the leg), geometry, ranges (hips ±30°, ankles 30–70°) and gear 150 after the MuJoCo model."""
    obs_dim, act_dim = 27, 8
    reset_noise, ctrl_range = 0.1, 1.0
    forward_weight, ctrl_cost, healthy_reward, z_range, vel_clip = 1.0, 0.5, 1.0, (0.2, 1.0), 50.0

    def __init__(self):
        density = 5.0                                        # the MuJoCo model's: a light body on strong, geared joints

        def capsule(length, radius):
            return density * (math.pi * radius * radius * length + 4.0 / 3.0 * math.pi * radius ** 3)

        r = 0.08
        leg = 0.2 * math.sqrt(2.0)
        torso_mass = density * 4.0 / 3.0 * math.pi * 0.25 ** 3 + 4 * capsule(leg, r)     # sphere + the four fixed stubs
        bodies = [Body("torso", -1, torso_mass, inertia=(0.4 * torso_mass * 0.25 ** 2 * 1.6,) * 3)]
        acts, contacts = [], [(0, (0.0, 0.0, -0.25))]
        for k, (sx, sy) in enumerate(((1, 1), (-1, 1), (-1, -1), (1, -1))):
            hip = (0.2 * sx, 0.2 * sy, 0.0)
            up = (0.2 * sx, 0.2 * sy, 0.0)                                                # upper segment, outward
            low = (0.4 * sx, 0.4 * sy, 0.0)                                               # lower segment, further outward
            m_up, m_low = capsule(leg, r), capsule(2 * leg, r)
            hip_id = len(bodies)
            bodies.append(Body(f"leg{k}_upper", 0, m_up, com=tuple(0.5 * v for v in up), inertia=_rod_inertia(m_up, up, r),
                               joint_pos=hip, axis=(0.0, 0.0, 1.0), range=(-math.radians(30), math.radians(30)),
                               damping=1.0, armature=1.0))
            # the ankle turns about the horizontal axis across the leg; a positive angle lowers the foot
            axis = (-sy / math.sqrt(2.0), sx / math.sqrt(2.0), 0.0)
            bodies.append(Body(f"leg{k}_lower", hip_id, m_low, com=tuple(0.5 * v for v in low),
                               inertia=_rod_inertia(m_low, low, r), joint_pos=up, axis=axis,
                               range=(math.radians(30), math.radians(70)), damping=1.0, armature=1.0))
            acts += [(hip_id, 150.0), (hip_id + 1, 150.0)]
            contacts += [(hip_id + 1, low), (hip_id + 1, (0.0, 0.0, 0.0))]
        self.engine = SpatialEngine(SpatialModel(bodies, actuators=acts, contacts=contacts, contact_stiffness=2.0e3,
                                                 contact_damping=3.0e1))
        self._theta0 = torch.tensor([0.0, math.radians(50)] * 4)
        self._pos0, self._pos_noise = torch.tensor([0.0, 0.0, 0.55]), torch.tensor([1.0, 1.0, 0.3])
        self._quat0 = torch.tensor([[1.0, 0.0, 0.0, 0.0]])

    def reset(self, key, batch, device):
        device = torch.device(device)
        k0, k1 = rnd.split(key.to(device))
        e = self.engine
        noise = self.reset_noise * (2.0 * rnd.uniform(k0, (batch, 3 + e.nj)) - 1.0)
        # (constants are copied to the device once: a host-to-device copy per call could not be recorded into a graph)
        pos = self._resident(self._pos0, device) + noise[:, :3] * self._resident(self._pos_noise, device)
        quat = self._resident(self._quat0, device).expand(batch, 4).contiguous()
        theta = self._resident(self._theta0, device) + noise[:, 3:]
        vel = self.reset_noise * rnd.normal(k1, (batch, e.ndof))
        st = {"pos": pos, "quat": quat, "theta": theta, "vel": vel}
        return st, self._observe(st)

    def _resident(self, t, device):
        key = (id(t), device)
        if key not in self.engine._resident:
            self.engine._resident[key] = (t, t.to(device))
        return self.engine._resident[key][1]

    def _observe(self, st):
        return torch.cat([st["pos"][:, 2:], st["quat"], st["theta"], st["vel"].clamp(-self.vel_clip, self.vel_clip)], dim=1)

    @property
    def frame_dt(self) -> float:
        return self.engine.model.dt * self.engine.model.substeps

    def step(self, env_state, action):
        a = action.to(env_state["pos"].dtype).clamp(-self.ctrl_range, self.ctrl_range)
        pos, quat, theta, vel = self.engine.step(env_state["pos"], env_state["quat"], env_state["theta"], env_state["vel"], a)
        finite = (torch.isfinite(pos).all(1) & torch.isfinite(quat).all(1) & torch.isfinite(theta).all(1)
                  & torch.isfinite(vel).all(1))
        keep = finite[:, None]
        pos, quat = torch.where(keep, pos, env_state["pos"]), torch.where(keep, quat, env_state["quat"])
        theta, vel = torch.where(keep, theta, env_state["theta"]), torch.where(keep, vel, torch.zeros_like(vel))
        healthy = (pos[:, 2] > self.z_range[0]) & (pos[:, 2] < self.z_range[1])
        vx = (pos[:, 0] - env_state["pos"][:, 0]) / self.frame_dt
        reward = self.forward_weight * vx + self.healthy_reward * healthy.to(vx.dtype) - self.ctrl_cost * (a * a).sum(1)
        reward = torch.where(finite, reward, torch.zeros_like(reward))
        st = {"pos": pos, "quat": quat, "theta": theta, "vel": vel}
        return st, self._observe(st), reward, ~healthy | ~finite

    def step_accumulate(self, env_state, action, total, done):
        """One step of a rollout with the bookkeeping of ``TensorEnvProblem.evaluate`` included (see
        ``PlanarEnv.step_accumulate``): on a GPU ONE launch (physics, observation, reward, termination, accumulation of
        the return: ``csrc/spatial.hip``), in place on the state and the accumulators it is given."""
        from ....ops import spatial as _k

        pos = env_state["pos"]
        if (_k.applies(self.engine, pos) and done.dtype == torch.bool
                and all(env_state[k].is_contiguous() for k in ("pos", "quat", "theta", "vel"))):
            task = dict(ctrl_range=self.ctrl_range, forward_weight=self.forward_weight, ctrl_cost=self.ctrl_cost,
                        healthy_reward=self.healthy_reward, z_range=self.z_range, vel_clip=self.vel_clip,
                        frame_dt=self.frame_dt)
            obs = _k.step_task(self.engine, task, pos, env_state["quat"], env_state["theta"], env_state["vel"], action,
                               self.obs_dim, total, done)
            return env_state, obs, total, done
        env_state, obs, reward, step_done = self.step(env_state, action)
        total = total + torch.where(done, torch.zeros_like(reward), reward.to(torch.float32))
        return env_state, obs, total, done | step_done

    def snapshot(self, env_state, index: int = 0):
        return self.engine.segments(env_state["pos"], env_state["quat"], env_state["theta"], index)


register_tensor_env("ant", Ant)
