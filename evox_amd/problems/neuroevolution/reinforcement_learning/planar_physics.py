"""Batched planar articulated-body physics on the device, and the locomotion / control tasks built on it.

Role parity: the environments the reference reaches through Brax (``src/evox/problems/neuroevolution/
reinforcement_learning/brax.py:15-133``).  Seven of Brax's tasks are planar mechanisms — ``inverted_pendulum``,
``inverted_double_pendulum``, ``reacher``, ``swimmer``, ``hopper``, ``walker2d``, ``halfcheetah`` — and are simulated
here with their observation / action sizes, reward and termination rules; geometry, masses, joint ranges and gears
follow the well-known MuJoCo model files of those tasks to within the simplifications below.  (``ant``, ``humanoid``
and the other spatial tasks need a 3-D engine and stay with the synthetic stand-ins of ``tensor_env.py``.)

The engine works in generalized coordinates ``q = (x, z, φ₀, θ₁ … θ_{L−1})`` of a tree of rigid rods in the x–z plane
(base position, absolute angle of the root link, relative joint angles), any of the three base coordinates may be
locked.  Every point of link i is ``base + Σ_k W[p, k]·u(α_k)`` with ``u`` the unit vectors of the links' absolute
angles and a STATIC weight matrix ``W`` (the kinematic tree folded into numbers), so positions, velocities, Jacobians
and the velocity-product accelerations are a handful of batched contractions:

    v_p = (ẋ, ż) + Σ_k W[p,k]·α̇_k·u⊥_k         ∂p/∂θ_j = Σ_k W[p,k]·Anc[k,j]·u⊥_k         a_p|q̈=0 = −Σ_k W[p,k]·α̇_k²·u_k

``M(q) q̈ = τ_act + τ_passive + Σ Jᵀ f_contact + Jᵀ m g − M-weighted velocity products`` is assembled and solved for all
environments at once (a batched Cholesky solve of order ≤ 9), and integrated by semi-implicit Euler in sub-steps.
Ground contact is a penalty force (spring + damper on penetration, smoothed Coulomb friction) at chosen points of the
links; joint limits are penalty torques; swimming uses resistive drag (normal ≫ tangential) on every link.

All of it is tensor code without host synchronisation: a rollout is a fixed sequence of launches and can be recorded
into a HIP graph together with the policy."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ....core import random as rnd
from .tensor_env import TensorEnv, register_tensor_env


# Dampers of the penalty forces fade in with the penetration instead of switching on at the surface: the forces are
# continuous in the state, so two computations that differ in the last bit do not part ways at a touch-down.
CONTACT_RAMP = 1.0e-3          # metres of penetration at which the contact damper is fully on
LIMIT_RAMP = 1.0e-2            # radians beyond a joint limit at which the limit damper is fully on


@dataclass
class Link:
    """A rigid rod.  ``joint_at``: distance from the rod's start to the joint that holds it (for the root: to the
    reference point whose coordinates are the base position).  ``attach``: where on the parent that joint sits,
    measured from the parent's start.  ``rest``: direction of the rod at zero joint angle, relative to the parent's
    direction (absolute for the root), radians."""
    name: str
    parent: int
    length: float
    mass: float
    rest: float = 0.0
    attach: float = 0.0
    joint_at: float = 0.0
    range: Optional[Tuple[float, float]] = None
    damping: float = 0.0
    stiffness: float = 0.0
    armature: float = 0.0
    inertia: Optional[float] = None              # about the centre; a uniform rod when None


@dataclass
class PlanarModel:
    links: List[Link]
    free_base: Tuple[bool, bool, bool] = (True, True, True)          # x, z, rotation of the root
    gravity: float = 9.81
    actuators: List[Tuple[int, float]] = field(default_factory=list)  # (dof index, gear); dof 0/1/2 = x/z/φ₀, 2+i = joint of link i
    contacts: List[Tuple[int, float]] = field(default_factory=list)   # (link, distance from the link's start)
    contact_stiffness: float = 5.0e4
    contact_damping: float = 5.0e2
    friction: float = 1.0
    base_damping: Tuple[float, float] = (0.0, 0.0)                    # on ẋ, ż (slider friction)
    drag: Tuple[float, float, float] = (0.0, 0.0, 0.0)                # tangential, normal (per metre of rod), angular
    limit_stiffness: float = 2.0e3
    limit_damping: float = 2.0e1
    dt: float = 0.002
    substeps: int = 4
    max_velocity: float = 0.25 / 0.002        # generalized velocities are clipped where a sub-step would turn a link by 0.25 rad


class PlanarEngine:
    """The static matrices of a :class:`PlanarModel` and one integration step for a batch of states ``(q, q̇)``."""

    def __init__(self, model: PlanarModel):
        self.model = model
        L = len(model.links)
        self.L, self.ndof = L, 2 + L
        anc = torch.zeros(L, L)
        for i, ln in enumerate(model.links):
            assert ln.parent < i, "links are listed parents first"
            k = i
            while k >= 0:
                anc[i, k] = 1.0
                k = model.links[k].parent
        self.anc = anc                                                # anc[i, k] = 1 when k is i or an ancestor of i
        rest = torch.tensor([ln.rest for ln in model.links])
        self.rest_abs = anc @ rest                                    # absolute direction at q = 0

        def weights(link: int, s: float) -> torch.Tensor:
            """Row of W for the point at distance s from the start of ``link``."""
            w = torch.zeros(L)
            w[link] = s - model.links[link].joint_at
            child = link
            while model.links[child].parent >= 0:
                par = model.links[child].parent
                w[par] += model.links[child].attach - model.links[par].joint_at
                child = par
            return w

        self.W_com = torch.stack([weights(i, 0.5 * ln.length) for i, ln in enumerate(model.links)])          # (L, L)
        self.W_con = (torch.stack([weights(i, s) for i, s in model.contacts]) if model.contacts
                      else torch.zeros(0, L))                                                                   # (C, L)
        self.mass = torch.tensor([ln.mass for ln in model.links])
        inertia = torch.tensor([ln.inertia if ln.inertia is not None else ln.mass * ln.length ** 2 / 12.0
                                for ln in model.links])
        # rotational part of the mass matrix and the armatures: constant
        rot = torch.zeros(self.ndof, self.ndof)
        rot[2:, 2:] = (anc * inertia[:, None]).T @ anc
        rot[2:, 2:] += torch.diag(torch.tensor([ln.armature for ln in model.links]))
        self.M_const = rot
        self.free = [d for d in range(self.ndof) if d >= 3 or model.free_base[d]]
        lo = torch.full((self.ndof,), -float("inf"))
        hi = torch.full((self.ndof,), float("inf"))
        for i, ln in enumerate(model.links):
            if ln.range is not None:
                lo[2 + i], hi[2 + i] = ln.range
        self.lo, self.hi = lo, hi
        self.damping = torch.tensor([model.base_damping[0], model.base_damping[1]] + [ln.damping for ln in model.links])
        self.stiffness = torch.tensor([0.0, 0.0] + [ln.stiffness for ln in model.links])
        gear = torch.zeros(len(model.actuators), self.ndof)
        for a, (dof, g) in enumerate(model.actuators):
            gear[a, dof] = g
        self.gear = gear                                              # (A, ndof)
        self.lengths = torch.tensor([ln.length for ln in model.links])
        self._dev: Dict = {}
        self._params: Dict = {}
        self._params_task: Dict = {}
        self._resident: Dict = {}

    def on(self, device):
        if device not in self._dev:
            names = ("anc", "rest_abs", "W_com", "W_con", "mass", "M_const", "lo", "hi", "damping", "stiffness", "gear",
                     "lengths")
            self._dev[device] = {n: getattr(self, n).to(device) for n in names}
            self._dev[device]["free"] = torch.tensor(self.free, device=device)
        return self._dev[device]

    # -- kinematics ------------------------------------------------------------------------------------------------
    def directions(self, q):
        """Absolute angles ``α`` (B, L) and unit vectors ``u`` (B, L, 2) of the links."""
        c = self.on(q.device)
        alpha = q[:, 2:] @ c["anc"].T + c["rest_abs"]
        return alpha, torch.stack([torch.cos(alpha), torch.sin(alpha)], dim=-1)

    def points(self, q, W):
        """Positions (B, P, 2) of the points with weight rows ``W`` (P, L)."""
        _, u = self.directions(q)
        return q[:, None, :2] + torch.einsum("pk,bkc->bpc", self.resident(W, q.device), u)

    def resident(self, t: torch.Tensor, device):
        """``t`` on ``device``, copied once (a copy from host memory per call would synchronise and cannot be recorded)."""
        if t.device == device:
            return t
        key = (id(t), device)
        if key not in self._resident:
            self._resident[key] = (t, t.to(device))               # the host tensor is kept so that its id stays taken
        return self._resident[key][1]

    def link_segments(self, q):
        """Start and end points of every link, (B, L, 2) each (for drawing)."""
        m = self.model
        starts = torch.stack([self._row(i, 0.0) for i in range(self.L)]).to(q.device)
        ends = torch.stack([self._row(i, m.links[i].length) for i in range(self.L)]).to(q.device)
        return self.points(q, starts), self.points(q, ends)

    def _row(self, link, s):
        w = self.W_com[link].clone()
        w[link] += s - 0.5 * self.model.links[link].length
        return w

    # -- dynamics --------------------------------------------------------------------------------------------------
    def acceleration(self, q, qd, ctrl):
        """``q̈`` (B, ndof; zero on locked coordinates) for controls ``ctrl`` (B, A) in [−1, 1]·range of the task."""
        m, c = self.model, self.on(q.device)
        B = q.shape[0]
        alpha, u = self.directions(q)
        up = torch.stack([-u[..., 1], u[..., 0]], dim=-1)                        # u⊥
        ad = qd[:, 2:] @ c["anc"].T                                              # α̇ (B, L)

        def jac(W):
            """Jacobian (B, P, 2, ndof) of the points with weights W."""
            P = W.shape[0]
            J = q.new_zeros(B, P, 2, self.ndof)
            J[:, :, 0, 0] = 1.0
            J[:, :, 1, 1] = 1.0
            J[:, :, :, 2:] = torch.einsum("pk,kj,bkc->bpcj", W, c["anc"], up)
            return J

        Jc = jac(c["W_com"])
        M = torch.einsum("p,bpcd,bpce->bde", c["mass"], Jc, Jc) + c["M_const"]
        a0 = -torch.einsum("pk,bk,bkc->bpc", c["W_com"], ad * ad, u)              # velocity products at the centres
        Q = -torch.einsum("p,bpcd,bpc->bd", c["mass"], Jc, a0)
        if m.gravity:
            Q = Q - m.gravity * torch.einsum("p,bpd->bd", c["mass"], Jc[:, :, 1, :])
        # passive joint forces and actuators
        Q = Q - c["damping"] * qd - c["stiffness"] * q
        below, above = (c["lo"] - q).clamp_min(0.0), (q - c["hi"]).clamp_min(0.0)
        outside = ((below + above) * (1.0 / LIMIT_RAMP)).clamp_max(1.0)          # the damper fades in over LIMIT_RAMP rad
        Q = Q + m.limit_stiffness * (below - above) - m.limit_damping * outside * qd
        if ctrl is not None and c["gear"].shape[0]:
            Q = Q + ctrl @ c["gear"]
        # ground contact
        if c["W_con"].shape[0]:
            Jp = jac(c["W_con"])
            pos_z = q[:, None, 1] + torch.einsum("pk,bk->bp", c["W_con"], u[..., 1])
            vel = torch.einsum("bpcd,bd->bpc", Jp, qd)
            depth = (-pos_z).clamp_min(0.0)
            touching = (depth * (1.0 / CONTACT_RAMP)).clamp_max(1.0)              # the damper fades in over CONTACT_RAMP m
            fn = (m.contact_stiffness * depth - m.contact_damping * vel[..., 1] * touching).clamp_min(0.0)
            ft = -m.friction * fn * torch.tanh(vel[..., 0] / 0.05)
            Q = Q + torch.einsum("bpcd,bpc->bd", Jp, torch.stack([ft, fn], dim=-1))
        # resistive drag on every link (swimming): force per link from the velocity of its centre, torque from its spin
        if any(m.drag):
            ct, cn, ca = m.drag
            vc = torch.einsum("bpcd,bd->bpc", Jc, qd)
            vt, vn = (vc * u).sum(-1), (vc * up).sum(-1)
            f = -(ct * vt[..., None] * u + cn * vn[..., None] * up) * c["lengths"][None, :, None]
            Q = Q + torch.einsum("bpcd,bpc->bd", Jc, f)
            Q[:, 2:] = Q[:, 2:] - ca * (ad * c["lengths"] ** 3 / 12.0) @ c["anc"]
        idx = c["free"]
        Mf = M[:, idx][:, :, idx]
        # (the unchecked factorisation: a state that has left the numbers gives NaNs here, which `PlanarEnv.step` ends
        # the episode on, instead of an exception and a host synchronisation per call)
        sol = torch.cholesky_solve(Q[:, idx, None], torch.linalg.cholesky_ex(Mf).L)[..., 0]
        qdd = torch.zeros_like(q)
        qdd[:, idx] = sol
        return qdd

    def step(self, q, qd, ctrl):
        """``substeps`` semi-implicit Euler steps of ``dt`` under constant controls.  fp32 states on a GPU take the
        one-lane-per-environment kernel (``csrc/planar.hip``: all sub-steps in one launch); the tensor formulation below
        is its oracle and the CPU path."""
        from ....ops import planar as _k

        if _k.applies(self, q):
            if q.device not in self._params:
                self._params[q.device] = _k.param_block(self).to(q.device)
            return _k.step(self, self._params[q.device], q, qd, ctrl)
        return self.step_tensor(q, qd, ctrl)

    def step_tensor(self, q, qd, ctrl):
        """The same step as batched tensor operations."""
        dt, vmax = self.model.dt, self.model.max_velocity
        for _ in range(self.model.substeps):
            qd = (qd + dt * self.acceleration(q, qd, ctrl)).clamp(-vmax, vmax)
            q = q + dt * qd
        return q, qd

    def energy(self, q, qd):
        """Kinetic + potential energy (B,) — conserved without actuation, damping, contact and limits."""
        c = self.on(q.device)
        alpha, u = self.directions(q)
        up = torch.stack([-u[..., 1], u[..., 0]], dim=-1)
        ad = qd[:, 2:] @ c["anc"].T
        vc = qd[:, None, :2] + torch.einsum("pk,bk,bkc->bpc", c["W_com"], ad, up)
        zc = q[:, None, 1] + torch.einsum("pk,bk->bp", c["W_com"], u[..., 1])
        rot = 0.5 * torch.einsum("bd,de,be->b", qd, c["M_const"], qd)
        return 0.5 * (c["mass"] * (vc * vc).sum(-1)).sum(-1) + rot + self.model.gravity * (c["mass"] * zc).sum(-1)


# ---------------------------------------------------------------------------------------------------------------------
# tasks
# ---------------------------------------------------------------------------------------------------------------------
def _capsule_mass(length: float, radius: float, density: float = 1000.0) -> float:
    return density * (math.pi * radius ** 2 * length + 4.0 / 3.0 * math.pi * radius ** 3)


class PlanarEnv(TensorEnv):
    """A task on a :class:`PlanarEngine`: ``reset`` draws the initial state around ``q0``, ``step`` clips the action,
    advances the physics and hands ``(q_before, q, q̇, action)`` to the task's ``_observe`` / ``_outcome``."""
    reset_noise = 0.005
    ctrl_range = 1.0

    def __init__(self, model: PlanarModel, q0: Sequence[float]):
        self.engine = PlanarEngine(model)
        self.q0 = torch.tensor(list(q0), dtype=torch.float32)
        assert self.q0.numel() == self.engine.ndof
        self.act_dim = len(model.actuators)

    def _noise_mask(self, device):
        if not hasattr(self, "_mask"):
            self._mask = torch.zeros(self.engine.ndof)
            self._mask[self.engine.free] = 1.0
        return self.engine.resident(self._mask, device)

    def reset(self, key, batch, device):
        device = torch.device(device)
        k0, k1, k2 = rnd.split(key.to(device), 3)
        mask = self._noise_mask(device)
        q = self.engine.resident(self.q0, device) + self.reset_noise * mask * (2.0 * rnd.uniform(k0, (batch, self.engine.ndof)) - 1.0)
        qd = self.reset_noise * mask * (2.0 * rnd.uniform(k1, (batch, self.engine.ndof)) - 1.0)
        extra = self._reset_extra(k2, batch, device)
        st = {"q": q, "qd": qd, **extra}
        return st, self._observe(st)

    def _reset_extra(self, key, batch, device):
        return {}

    def step(self, env_state, action):
        q0, qd0 = env_state["q"], env_state["qd"]
        a = action.to(q0.dtype).clamp(-self.ctrl_range, self.ctrl_range)
        q, qd = self.engine.step(q0, qd0, a)
        st = dict(env_state, q=q, qd=qd)
        reward, done = self._outcome(env_state, st, a)
        # a state that left the numbers (a policy that drives the penalty contacts unstable) ends the episode
        bad = ~(torch.isfinite(q).all(1) & torch.isfinite(qd).all(1))
        st["q"] = torch.where(bad[:, None], q0, q)
        st["qd"] = torch.where(bad[:, None], torch.zeros_like(qd), qd)
        reward = torch.where(bad, torch.zeros_like(reward), reward)
        return st, self._observe(st), reward, done | bad

    def _recipe(self):
        """What ``csrc/planar.hip`` needs to apply this task's observation, reward and termination in the step kernel
        (see ``ops.planar.param_block``), or None when the task has no recipe there."""
        return None

    _aux_key = None                  # name of the per-environment data the recipe reads (the reacher's targets)

    def step_accumulate(self, env_state, action, total, done):
        """One step of a rollout, with the bookkeeping of ``TensorEnvProblem.evaluate`` included: returns
        ``(env_state, obs, total, done)`` with the step's reward added to ``total`` where ``done`` was not yet set.  On a
        GPU this is ONE launch for a task with a recipe (physics, observation, reward, termination, accumulation:
        ``csrc/planar.hip``), and it works IN PLACE on the state and the accumulators it is given (a rollout owns them)."""
        from ....ops import planar as _k

        q = env_state["q"]
        recipe = self._recipe()
        if (recipe is not None and _k.applies(self.engine, q) and q.is_contiguous() and env_state["qd"].is_contiguous()
                and done.dtype == torch.bool):
            eng = self.engine
            if q.device not in eng._params_task:
                eng._params_task[q.device] = _k.param_block(eng, recipe).to(q.device)
            aux = env_state[self._aux_key] if self._aux_key else None
            obs = _k.step_task(eng, eng._params_task[q.device], recipe["recipe"], q, env_state["qd"], action, self.obs_dim,
                               total, done, aux)
            return env_state, obs, total, done
        env_state, obs, reward, step_done = self.step(env_state, action)
        total = total + torch.where(done, torch.zeros_like(reward), reward.to(torch.float32))
        return env_state, obs, total, done | step_done

    # snapshots for drawing: the link segments of one environment
    def snapshot(self, env_state, index: int = 0):
        s, e = self.engine.link_segments(env_state["q"][index:index + 1])
        return torch.stack([s[0], e[0]], dim=1).detach().cpu()            # (L, 2, 2)

    @property
    def frame_dt(self) -> float:
        return self.engine.model.dt * self.engine.model.substeps


class _Locomotion(PlanarEnv):
    """Forward-velocity reward, control cost, healthy bonus; observation = q without x, then q̇ (clipped)."""
    forward_weight, ctrl_cost, healthy_reward = 1.0, 1e-3, 1.0
    z_range: Optional[Tuple[float, float]] = None
    angle_range: Optional[Tuple[float, float]] = None
    terminate = True
    vel_clip = 10.0

    def _observe(self, st):
        return torch.cat([st["q"][:, 1:], st["qd"].clamp(-self.vel_clip, self.vel_clip)], dim=1)

    obs_skip = 1                     # leading coordinates left out of the observation (x; the swimmer also drops z)

    def _recipe(self):
        return dict(recipe=1, ctrl_range=self.ctrl_range, forward_weight=self.forward_weight, ctrl_cost=self.ctrl_cost,
                    healthy_reward=self.healthy_reward, terminate=self.terminate, z_range=self.z_range,
                    angle_range=self.angle_range, obs_skip=self.obs_skip, vel_clip=self.vel_clip, frame_dt=self.frame_dt)

    def _healthy(self, st):
        ok = torch.ones(st["q"].shape[0], dtype=torch.bool, device=st["q"].device)
        if self.z_range is not None:
            ok = ok & (st["q"][:, 1] > self.z_range[0]) & (st["q"][:, 1] < self.z_range[1])
        if self.angle_range is not None:
            ok = ok & (st["q"][:, 2] > self.angle_range[0]) & (st["q"][:, 2] < self.angle_range[1])
        return ok

    def _outcome(self, before, after, a):
        vx = (after["q"][:, 0] - before["q"][:, 0]) / self.frame_dt
        healthy = self._healthy(after)
        bonus = self.healthy_reward * (healthy.to(vx.dtype) if self.terminate else torch.ones_like(vx))
        reward = self.forward_weight * vx + bonus - self.ctrl_cost * (a * a).sum(1)
        return reward, (~healthy if self.terminate else torch.zeros_like(healthy))


def _leg(links, prefix, parent, attach, thigh, leg, foot, foot_back, ranges, radii, damping=1.0, armature=1.0,
         rest=(math.pi, 0.0, math.pi / 2)):
    """thigh – leg – foot below ``parent``; returns the indices of the three links.  ``armature`` (rotor inertia
    reflected to the joint, as in the MuJoCo models) also keeps the light feet from chattering on the stiff contacts."""
    base = len(links)
    links.append(Link(prefix + "thigh", parent, thigh, _capsule_mass(thigh, radii[0]), rest=rest[0], attach=attach,
                      range=ranges[0], damping=damping, armature=armature))
    links.append(Link(prefix + "leg", base, leg, _capsule_mass(leg, radii[1]), rest=rest[1], attach=thigh, range=ranges[1],
                      damping=damping, armature=armature))
    links.append(Link(prefix + "foot", base + 1, foot, _capsule_mass(foot, radii[2]), rest=rest[2], attach=leg,
                      joint_at=foot_back, range=ranges[2], damping=damping, armature=armature))
    return base, base + 1, base + 2


_D = math.pi / 180.0


class Hopper(_Locomotion):
    """One-legged hopper: torso, thigh, leg, foot (observation 11, action 3); ends when the torso drops below 0.7 m or
    leans by more than 0.2 rad."""
    obs_dim, z_range, angle_range = 11, (0.7, float("inf")), (-0.2, 0.2)

    def __init__(self):
        links = [Link("torso", -1, 0.4, _capsule_mass(0.4, 0.05), rest=math.pi / 2, joint_at=0.2)]
        t, l, f = _leg(links, "", 0, 0.0, 0.45, 0.5, 0.39, 0.13, [(-150 * _D, 0.0), (-150 * _D, 0.0), (-45 * _D, 45 * _D)],
                       (0.05, 0.04, 0.06))
        model = PlanarModel(links, actuators=[(2 + t, 200.0), (2 + l, 200.0), (2 + f, 200.0)],
                            contacts=[(f, 0.0), (f, 0.39), (l, 0.0), (t, 0.0), (0, 0.4)], friction=2.0)
        super().__init__(model, [0.0, 1.25, 0.0, 0.0, 0.0, 0.0])


class Walker2d(_Locomotion):
    """Two-legged walker (observation 17, action 6); ends when the torso leaves 0.8–2.0 m or leans by more than 1 rad."""
    obs_dim, z_range, angle_range = 17, (0.8, 2.0), (-1.0, 1.0)

    def __init__(self):
        links = [Link("torso", -1, 0.4, _capsule_mass(0.4, 0.05), rest=math.pi / 2, joint_at=0.2)]
        ranges = [(-150 * _D, 0.0), (-150 * _D, 0.0), (-45 * _D, 45 * _D)]
        right = _leg(links, "right_", 0, 0.0, 0.45, 0.5, 0.2, 0.0, ranges, (0.05, 0.04, 0.06), damping=0.1, armature=0.05)
        left = _leg(links, "left_", 0, 0.0, 0.45, 0.5, 0.2, 0.0, ranges, (0.05, 0.04, 0.06), damping=0.1, armature=0.05)
        acts = [(2 + right[0], 100.0), (2 + right[1], 100.0), (2 + right[2], 20.0),
                (2 + left[0], 100.0), (2 + left[1], 100.0), (2 + left[2], 20.0)]
        contacts = [(right[2], 0.0), (right[2], 0.2), (left[2], 0.0), (left[2], 0.2), (right[1], 0.0), (left[1], 0.0),
                    (right[0], 0.0), (0, 0.4)]
        model = PlanarModel(links, actuators=acts, contacts=contacts, friction=0.9)
        super().__init__(model, [0.0, 1.25, 0.0] + [0.0] * 6)


class HalfCheetah(_Locomotion):
    """Planar cheetah: torso with a back and a front leg of three links each, spring-loaded joints (observation 17,
    action 6); never terminates; control cost 0.1."""
    obs_dim, ctrl_cost, terminate, healthy_reward = 17, 0.1, False, 0.0
    reset_noise = 0.1

    def __init__(self):
        r = 0.046
        links = [Link("torso", -1, 1.0, _capsule_mass(1.0, r) + _capsule_mass(0.3, r), rest=0.0, joint_at=0.5)]

        def chain(prefix, attach, lengths, angles, ranges, stiff, damp):
            idx, parent, prev_abs, at = [], 0, 0.0, attach
            for n, (ln, ang, rg, k, d) in enumerate(zip(lengths, angles, ranges, stiff, damp)):
                links.append(Link(prefix + ("thigh", "shin", "foot")[n], parent, ln, _capsule_mass(ln, r), rest=ang - prev_abs,
                                  attach=at, range=rg, stiffness=k, damping=d, armature=0.1))
                parent, prev_abs, at = len(links) - 1, ang, ln
                idx.append(parent)
            return idx

        back = chain("b", 0.0, (0.29, 0.30, 0.188), (-52 * _D, -153 * _D, -73 * _D),
                     [(-0.52, 1.05), (-0.785, 0.785), (-0.4, 0.785)], (240.0, 180.0, 120.0), (6.0, 4.5, 3.0))
        front = chain("f", 1.0, (0.266, 0.212, 0.14), (-120 * _D, -54 * _D, -57 * _D),
                      [(-1.0, 0.7), (-1.2, 0.87), (-0.5, 0.5)], (180.0, 120.0, 60.0), (4.5, 3.0, 1.5))
        acts = [(2 + i, g) for i, g in zip(back + front, (120.0, 90.0, 60.0, 120.0, 60.0, 30.0))]
        contacts = [(back[2], 0.188), (front[2], 0.14), (back[1], 0.30), (front[1], 0.212), (back[0], 0.29), (front[0], 0.266),
                    (0, 0.0), (0, 1.0)]
        model = PlanarModel(links, actuators=acts, contacts=contacts, friction=0.4)
        super().__init__(model, [0.0, 0.7, 0.0] + [0.0] * 6)


class Swimmer(_Locomotion):
    """Three-link swimmer in a viscous fluid, no gravity (observation 8 = the three angles and five velocities, action 2):
    forward velocity minus 1e-4 of the squared controls."""
    obs_dim, ctrl_cost, terminate, healthy_reward, obs_skip = 8, 1e-4, False, 0.0, 2
    reset_noise = 0.1

    def __init__(self):
        m = _capsule_mass(1.0, 0.1)
        links = [Link("torso", -1, 1.0, m, rest=math.pi, joint_at=0.0),
                 Link("mid", 0, 1.0, m, attach=1.0, range=(-100 * _D, 100 * _D)),
                 Link("back", 1, 1.0, m, attach=1.0, range=(-100 * _D, 100 * _D))]
        model = PlanarModel(links, gravity=0.0, actuators=[(3, 150.0), (4, 150.0)], drag=(2.0, 40.0, 40.0), dt=0.005, substeps=4,
                            max_velocity=50.0)
        super().__init__(model, [0.0, 0.0, 0.0, 0.0, 0.0])

    def _observe(self, st):
        return torch.cat([st["q"][:, 2:], st["qd"].clamp(-self.vel_clip, self.vel_clip)], dim=1)


class InvertedPendulum(PlanarEnv):
    """Cart on a rail with a free pole (observation 4 = x, θ, ẋ, θ̇; action 1 in [−3, 3]): one point per step while
    |θ| ≤ 0.2 rad."""
    obs_dim, ctrl_range, reset_noise = 4, 3.0, 0.01

    def __init__(self):
        links = [Link("cart", -1, 0.0, 10.47, inertia=0.0),
                 Link("pole", 0, 0.6, _capsule_mass(0.6, 0.049), rest=math.pi / 2, range=(-90 * _D, 90 * _D))]
        links[1].damping = 1.0
        model = PlanarModel(links, free_base=(True, False, False), actuators=[(0, 100.0)], base_damping=(1.0, 0.0),
                            dt=0.01, substeps=2, max_velocity=25.0)
        super().__init__(model, [0.0, 0.0, 0.0, 0.0])

    def _recipe(self):
        return dict(recipe=2, ctrl_range=self.ctrl_range, max_angle=0.2)

    def _observe(self, st):
        return torch.stack([st["q"][:, 0], st["q"][:, 3], st["qd"][:, 0], st["qd"][:, 3]], dim=1)

    def _outcome(self, before, after, a):
        ok = after["q"][:, 3].abs() <= 0.2
        return ok.to(after["q"].dtype), ~ok


class InvertedDoublePendulum(PlanarEnv):
    """Cart with two poles in series (observation 8 = x, sin θ₁₂, cos θ₁₂, clipped velocities; action 1): 10 per step
    minus a distance and a velocity penalty while the tip stays above 1 m."""
    obs_dim, ctrl_range, reset_noise = 8, 1.0, 0.05

    def __init__(self):
        pm = _capsule_mass(0.6, 0.045)
        links = [Link("cart", -1, 0.0, 10.47, inertia=0.0),
                 Link("pole", 0, 0.6, pm, rest=math.pi / 2, damping=0.05),
                 Link("pole2", 1, 0.6, pm, attach=0.6, range=(-1.0, 1.0), damping=0.05)]
        model = PlanarModel(links, free_base=(True, False, False), actuators=[(0, 500.0)], base_damping=(0.05, 0.0),
                            dt=0.005, substeps=4, max_velocity=50.0)
        super().__init__(model, [0.0] * 5)
        self._tip = self.engine._row(2, 0.6)[None]

    def _recipe(self):
        return dict(recipe=3, ctrl_range=self.ctrl_range, tip=self._tip)

    def _observe(self, st):
        q, qd = st["q"], st["qd"]
        th = q[:, 3:5]
        # (slices, not an index list: that would be a host-to-device copy per call, which cannot be recorded)
        return torch.cat([q[:, :1], torch.sin(th), torch.cos(th), qd[:, :1].clamp(-10, 10), qd[:, 3:5].clamp(-10, 10)], dim=1)

    def _outcome(self, before, after, a):
        tip = self.engine.points(after["q"], self._tip)[:, 0]
        dist = 0.01 * tip[:, 0] ** 2 + (tip[:, 1] - 1.2) ** 2
        vel = 1e-3 * after["qd"][:, 3] ** 2 + 5e-3 * after["qd"][:, 4] ** 2
        return 10.0 - dist - vel, tip[:, 1] <= 1.0


class Reacher(PlanarEnv):
    """Two-link arm on a fixed pivot reaching for a random target, no gravity (observation 11 = cos θ, sin θ, target, θ̇,
    fingertip − target with a zero third coordinate; action 2): minus the distance minus the squared controls."""
    obs_dim, reset_noise = 11, 0.1

    def __init__(self):
        links = [Link("link0", -1, 0.1, _capsule_mass(0.1, 0.01), armature=1.0, damping=1.0),
                 Link("link1", 0, 0.11, _capsule_mass(0.11, 0.01), attach=0.1, armature=1.0, damping=1.0, range=(-3.0, 3.0))]
        model = PlanarModel(links, free_base=(False, False, True), gravity=0.0, actuators=[(2, 200.0), (3, 200.0)],
                            dt=0.01, substeps=2, max_velocity=25.0)
        super().__init__(model, [0.0] * 4)
        self._tip = self.engine._row(1, 0.11)[None]

    _aux_key = "target"

    def _recipe(self):
        return dict(recipe=4, ctrl_range=self.ctrl_range, tip=self._tip)

    def _reset_extra(self, key, batch, device):
        # uniform in the disk of radius 0.2
        ku, ka = rnd.split(key)
        r = 0.2 * torch.sqrt(rnd.uniform(ku, (batch,)))
        ang = 2.0 * math.pi * rnd.uniform(ka, (batch,))
        return {"target": torch.stack([r * torch.cos(ang), r * torch.sin(ang)], dim=1)}

    def _observe(self, st):
        q, qd = st["q"], st["qd"]
        th = q[:, 2:4]
        to_target = self.engine.points(q, self._tip)[:, 0] - st["target"]
        return torch.cat([torch.cos(th), torch.sin(th), st["target"], qd[:, 2:4], to_target, torch.zeros_like(to_target[:, :1])], dim=1)

    def _outcome(self, before, after, a):
        dist = (self.engine.points(after["q"], self._tip)[:, 0] - after["target"]).norm(dim=1)
        done = torch.zeros_like(dist, dtype=torch.bool)
        return -dist - (a * a).sum(1), done


PLANAR_TASKS = {"inverted_pendulum": InvertedPendulum, "inverted_double_pendulum": InvertedDoublePendulum,
                "reacher": Reacher, "swimmer": Swimmer, "hopper": Hopper, "walker2d": Walker2d, "halfcheetah": HalfCheetah}
for _name, _cls in PLANAR_TASKS.items():
    register_tensor_env(_name, _cls)
