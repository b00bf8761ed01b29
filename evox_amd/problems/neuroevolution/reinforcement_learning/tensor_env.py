"""Rollouts of batched on-device environments.
Role parity: reference ``src/evox/problems/neuroevolution/reinforcement_learning/brax.py:15-133``
(Brax is a JAX physics engine and cannot run here; the rollout problem is written against a
small tensor-environment protocol instead, and ships a synthetic locomotion-style environment
with Brax-like observation/action sizes).

The whole rollout – policy forward of every individual, environment step, reward
accumulation, done masking – stays on the device; the loop over time steps issues a fixed,
shape-static sequence of kernels and is therefore HIP-graph capturable."""
from __future__ import annotations

import math
from typing import Callable, Dict

import torch

from ....core import random as rnd
from ....core.problem import Problem
from ....core.state import State


class TensorEnv:
    """Protocol of a batched environment living on one device.

    reset(key, batch)        -> (env_state, obs[batch, obs_dim])
    step(env_state, action)  -> (env_state, obs, reward[batch], done[batch] (bool))
    """
    obs_dim: int
    act_dim: int

    def reset(self, key, batch, device):
        raise NotImplementedError

    def step(self, env_state, action):
        raise NotImplementedError


class SyntheticLocomotion(TensorEnv):
    """A synthetic stand-in for a locomotion task: a randomly drawn, stable, lightly non-linear
    system ``s' = tanh(A s + B a)·decay + drift``; the reward pays for velocity along a fixed
    direction and charges a control cost; the episode ends when the 'torso height' coordinate
    leaves its healthy range.  Sizes default to those of Brax's ``ant`` (obs 27, act 8)."""

    def __init__(self, obs_dim=27, act_dim=8, seed=0, ctrl_cost=0.05, healthy_range=(-2.5, 2.5), noise=0.01):
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.ctrl_cost, self.healthy_range, self.noise = ctrl_cost, healthy_range, noise
        g = torch.Generator().manual_seed(seed)
        A = torch.randn(obs_dim, obs_dim, generator=g) / math.sqrt(obs_dim)
        self.A = 0.9 * A / torch.linalg.matrix_norm(A, ord=2)
        self.B = torch.randn(act_dim, obs_dim, generator=g) / math.sqrt(act_dim)
        self.forward_dir = torch.nn.functional.normalize(torch.randn(obs_dim, generator=g), dim=0)
        self._dev = {}

    def _on(self, device):
        if device not in self._dev:
            self._dev[device] = (self.A.T.contiguous().to(device), self.B.to(device), self.forward_dir.to(device))
        return self._dev[device]

    def reset(self, key, batch, device):
        k0, k1 = rnd.split(key.to(device))
        s = 0.1 * rnd.normal(k0, (batch, self.obs_dim))
        return {"s": s, "key": k1, "t": 0}, s

    def step(self, env_state, action):
        s = env_state["s"]
        At, B, fwd = self._on(s.device)
        a = action.to(s.dtype).clamp(-1, 1)
        key = rnd.fold_in(env_state["key"], env_state["t"])
        nxt = torch.tanh(torch.addmm(a @ B, s, At)) + self.noise * rnd.normal(key, s.shape)
        reward = (nxt - s) @ fwd * 10.0 - self.ctrl_cost * (a * a).sum(1) + 0.05
        lo, hi = self.healthy_range
        done = (nxt[:, 0] < lo) | (nxt[:, 0] > hi)
        return {"s": nxt, "key": env_state["key"], "t": env_state["t"] + 1}, nxt, reward, done


_REGISTRY: Dict[str, Callable[[], TensorEnv]] = {
    # observation / action sizes of the Brax tasks of the same name
    "synthetic-ant": lambda: SyntheticLocomotion(27, 8, seed=1),
    "synthetic-halfcheetah": lambda: SyntheticLocomotion(17, 6, seed=2),
    "synthetic-hopper": lambda: SyntheticLocomotion(11, 3, seed=3),
    "synthetic-humanoid": lambda: SyntheticLocomotion(244, 17, seed=4),
    "synthetic-swimmer": lambda: SyntheticLocomotion(8, 2, seed=5),
    "synthetic-walker2d": lambda: SyntheticLocomotion(17, 6, seed=6),
}


def register_tensor_env(name: str, factory: Callable[[], TensorEnv]):
    _REGISTRY[name] = factory


def make_tensor_env(name: str) -> TensorEnv:
    if name not in _REGISTRY:
        raise KeyError(f"unknown tensor environment {name!r}; registered: {sorted(_REGISTRY)}")
    return _REGISTRY[name]()


class TensorEnvProblem(Problem):
    """Fitness of a population of policies = (reduced) episode return over ``num_episodes``
    rollouts of at most ``max_episode_length`` steps.

    policy(weights, obs): batched over the population – ``weights`` is whatever the workflow's
    candidate transforms produce with leading dimension ``pop``, ``obs`` is
    ``(pop, num_episodes, obs_dim)``; returns ``(pop, num_episodes, act_dim)``.  (The reference
    takes a per-individual policy and vmaps it twice, ``brax.py:45``; here batching is the
    policy's job so that it can map to batched GEMMs – see ``evox_amd.models.MLP``.)

    Returns are rewards, so pair it with ``opt_direction="max"`` in the workflow."""

    def __init__(self, policy: Callable, env: TensorEnv, max_episode_length: int, num_episodes: int,
                 reduce_fn: Callable = torch.mean, early_exit_every: int = 0):
        super().__init__()
        self.policy = policy
        self.env = env
        self.max_episode_length = max_episode_length
        self.num_episodes = num_episodes
        self.reduce_fn = reduce_fn
        # 0: never read the done flags back (no host sync at all); k>0: check every k steps
        self.early_exit_every = early_exit_every

    def setup(self, key):
        return State(key=key)

    def evaluate(self, state, weights):
        from ....core.tree import tree_leaves
        lead = tree_leaves(weights)[0]
        pop, device = lead.shape[0], lead.device
        E = self.num_episodes
        key, eval_key = rnd.split(state.key.to(device))
        env_state, obs = self.env.reset(eval_key, pop * E, device)
        done = torch.zeros(pop * E, dtype=torch.bool, device=device)
        total = torch.zeros(pop * E, dtype=torch.float32, device=device)
        # an environment may take the bookkeeping of the loop into its own step (one launch instead of a dozen)
        accumulate = getattr(self.env, "step_accumulate", None)
        for t in range(self.max_episode_length):
            action = self.policy(weights, obs.reshape(pop, E, -1)).reshape(pop * E, -1)
            if accumulate is not None:
                env_state, obs, total, done = accumulate(env_state, action, total, done)
            else:
                env_state, obs, reward, step_done = self.env.step(env_state, action)
                total = total + torch.where(done, torch.zeros_like(reward), reward.to(torch.float32))
                done = done | step_done
            if self.early_exit_every and (t + 1) % self.early_exit_every == 0 and bool(done.all()):
                break
        total = self.reduce_fn(total.reshape(pop, E), dim=-1)
        if isinstance(total, tuple):         # torch.max / torch.median return (values, indices)
            total = total[0]
        return total, state.replace(key=key)

    # ------------------------------------------------------------------------------------
    def rollout(self, key, weights, respect_done: bool = False, snapshots: bool = False):
        """One episode of ONE individual (``weights`` without the population axis): the observations
        ``(T+1, obs_dim)``, rewards ``(T,)`` and done flags ``(T,)`` on the host.  ``snapshots=True`` adds a fourth
        item: what the environment's ``snapshot(env_state)`` returned at every time step (the link segments of a planar
        mechanism), or ``None`` when the environment has no such method."""
        from ....core.tree import tree_leaves, tree_map
        device = tree_leaves(weights)[0].device
        batched = tree_map(lambda w: w.unsqueeze(0), weights)
        env_state, obs = self.env.reset(key.to(device), 1, device)
        snap = getattr(self.env, "snapshot", None) if snapshots else None
        frames, rewards, dones = [obs[0].detach().cpu()], [], []
        shots = [snap(env_state)] if snap is not None else None
        for _ in range(self.max_episode_length):
            action = self.policy(batched, obs.reshape(1, 1, -1)).reshape(1, -1)
            env_state, obs, reward, done = self.env.step(env_state, action)
            frames.append(obs[0].detach().cpu())
            rewards.append(float(reward[0]))
            dones.append(bool(done[0]))
            if shots is not None:
                shots.append(snap(env_state))
            if respect_done and dones[-1]:
                break
        out = (torch.stack(frames), torch.tensor(rewards), torch.tensor(dones))
        return out + (shots,) if snapshots else out

    def visualize(self, key, weights, output_type: str = "HTML", respect_done: bool = False, width: int = 320,
                  height: int = 240, **kwargs):
        """Roll one individual out and render the episode (reference: ``brax.py:99-133``, which renders the physics
        scene through ``brax.io``).  A planar mechanism (an environment with ``snapshot``) is drawn as its links over the
        ground line, the camera following it; any other tensor environment has no scene, so what is drawn is its
        observation vector.

        ``output_type="HTML"``: a self-contained page — the animated scene (or bar display of the observation) with a
        slider, and the cumulative reward curve.  ``"rgb_array"``: one ``(height, width, 3)`` uint8 array per time step.
        An environment may supply its own ``render(obs_row, width, height) -> uint8 array`` for the frames."""
        assert output_type in ("HTML", "rgb_array"), "output_type must be either HTML or rgb_array"
        obs, rewards, dones, shots = self.rollout(key, weights, respect_done, snapshots=True)
        if output_type == "rgb_array":
            if shots is not None:
                return [_links_frame(seg, width, height) for seg in shots]
            draw = getattr(self.env, "render", None) or _bars_frame
            return [draw(row, width, height) for row in obs]
        import json
        lim = float(obs.abs().max().clamp_min(1e-6))
        payload = {"obs": [[round(float(v), 5) for v in row] for row in obs],
                   "ret": [round(float(v), 5) for v in torch.cumsum(rewards, 0)],
                   "done": [int(d) for d in dones], "lim": lim}
        if shots is not None:
            payload["seg"] = [[[round(float(v), 4) for v in link.reshape(-1)] for link in seg] for seg in shots]
        return _HTML_TEMPLATE.replace("__DATA__", json.dumps(payload)).replace("__W__", str(width)).replace("__H__", str(height))


def _links_frame(segments, width: int, height: int, metres: float = 3.0):
    """Link segments (L, 2, 2) = (link, start/end, x/z) as dark lines over a ground line; ``metres`` of scene height,
    the camera centred on the mechanism horizontally."""
    import numpy as np

    img = np.full((height, width, 3), 255, dtype=np.uint8)
    seg = np.asarray(segments, dtype=np.float64)
    scale = height / metres
    cx = float(seg[..., 0].mean())
    ground = int(round(height * 0.85))

    def px(x, z):
        return (x - cx) * scale + width / 2.0, ground - z * scale

    img[min(max(ground, 0), height - 1), :, :] = 120
    # distance marks on the ground every half metre, so that motion shows although the camera follows
    first = math.floor((cx - width / 2.0 / scale) * 2.0) / 2.0
    for k in range(int(width / scale * 2) + 3):
        u, _ = px(first + 0.5 * k, 0.0)
        if 0 <= int(u) < width:
            img[ground:min(ground + 4, height), int(u), :] = 120
    for (x0, z0), (x1, z1) in seg:
        (u0, v0), (u1, v1) = px(x0, z0), px(x1, z1)
        n = int(max(abs(u1 - u0), abs(v1 - v0))) + 2
        us = np.clip(np.round(np.linspace(u0, u1, n)).astype(int), 0, width - 1)
        vs = np.clip(np.round(np.linspace(v0, v1, n)).astype(int), 0, height - 1)
        for du in (-1, 0, 1):
            for dv in (-1, 0, 1):
                img[np.clip(vs + dv, 0, height - 1), np.clip(us + du, 0, width - 1)] = (30, 60, 140)
    return img


def _bars_frame(row, width: int, height: int):
    """Observation vector as a bar display: one column per coordinate, zero line in the middle."""
    import numpy as np

    img = np.full((height, width, 3), 255, dtype=np.uint8)
    v = np.asarray(row, dtype=np.float64)
    n = max(len(v), 1)
    lim = max(float(np.abs(v).max()), 1.0)
    mid = height // 2
    img[mid, :, :] = 160
    bw = max(width // n, 1)
    for i, x in enumerate(v[: width // bw]):
        h = int(round((height // 2 - 2) * min(abs(x) / lim, 1.0)))
        lo, hi = (mid - h, mid) if x >= 0 else (mid, mid + h)
        img[lo:hi + 1, i * bw:(i + 1) * bw - (1 if bw > 2 else 0)] = (38, 110, 200) if x >= 0 else (200, 80, 50)
    return img


_HTML_TEMPLATE = """<!DOCTYPE html>
<html><head><meta charset="utf-8"><title>episode</title></head>
<body style="font-family:sans-serif">
<canvas id="c" width="__W__" height="__H__" style="border:1px solid #999"></canvas>
<canvas id="r" width="__W__" height="__H__" style="border:1px solid #999"></canvas><br>
<input id="t" type="range" min="0" value="0" style="width:640px"> <span id="lab"></span>
<script>
const D = __DATA__;
const c = document.getElementById('c').getContext('2d'), r = document.getElementById('r').getContext('2d');
const W = __W__, H = __H__, T = D.obs.length, sl = document.getElementById('t');
sl.max = T - 1;
function scene(t) {
  const S = D.seg[t], sc = H / 3.0, g = H * 0.85;
  let cx = 0; S.forEach(l => { cx += (l[0] + l[2]) / 2; }); cx /= S.length;
  c.strokeStyle = '#777'; c.lineWidth = 1; c.beginPath(); c.moveTo(0, g); c.lineTo(W, g); c.stroke();
  const first = Math.floor((cx - W / 2 / sc) * 2) / 2;
  for (let k = 0; k < W / sc * 2 + 3; k++) { const u = (first + 0.5 * k - cx) * sc + W / 2; c.beginPath(); c.moveTo(u, g); c.lineTo(u, g + 4); c.stroke(); }
  c.strokeStyle = '#1e3c8c'; c.lineWidth = 3; c.lineCap = 'round';
  S.forEach(l => { c.beginPath(); c.moveTo((l[0] - cx) * sc + W / 2, g - l[1] * sc); c.lineTo((l[2] - cx) * sc + W / 2, g - l[3] * sc); c.stroke(); });
}
function draw(t) {
  c.clearRect(0, 0, W, H);
  if (D.seg) { scene(t); } else {
  c.strokeStyle = '#999'; c.beginPath(); c.moveTo(0, H / 2); c.lineTo(W, H / 2); c.stroke();
  const o = D.obs[t], bw = W / o.length;
  for (let i = 0; i < o.length; i++) {
    const h = (H / 2 - 2) * Math.min(Math.abs(o[i]) / D.lim, 1);
    c.fillStyle = o[i] >= 0 ? '#266ec8' : '#c85032';
    c.fillRect(i * bw, o[i] >= 0 ? H / 2 - h : H / 2, Math.max(bw - 1, 1), h);
  }
  }
  r.clearRect(0, 0, W, H);
  if (D.ret.length) {
    const lo = Math.min(0, ...D.ret), hi = Math.max(1e-9, ...D.ret);
    r.strokeStyle = '#2a2'; r.beginPath();
    D.ret.forEach((v, i) => { const x = i / Math.max(D.ret.length - 1, 1) * W, y = H - (v - lo) / (hi - lo) * (H - 4) - 2;
      i ? r.lineTo(x, y) : r.moveTo(x, y); });
    r.stroke();
    const x = Math.max(t - 1, 0) / Math.max(D.ret.length - 1, 1) * W;
    r.strokeStyle = '#000'; r.beginPath(); r.moveTo(x, 0); r.lineTo(x, H); r.stroke();
  }
  document.getElementById('lab').textContent = 'step ' + t + '/' + (T - 1) + (t > 0 ? '  return ' + D.ret[t - 1] : '') +
    (t > 0 && D.done[t - 1] ? '  (done)' : '');
}
sl.oninput = () => draw(+sl.value);
let t = 0; draw(0);
const timer = setInterval(() => { t = (t + 1) % T; sl.value = t; draw(t); }, 50);
sl.onmousedown = () => clearInterval(timer);
</script></body></html>
"""


class Brax(TensorEnvProblem):
    """Name-based constructor with the reference's signature (``brax.py:16-24``).  ``env_name``
    is looked up in the tensor-environment registry (``register_tensor_env``); the real Brax
    engine needs JAX and is not available to this framework."""

    def __init__(self, policy, env_name, max_episode_length, num_episodes, reduce_fn=torch.mean, backend="generalized"):
        self.env_name, self.backend = env_name, backend
        super().__init__(policy, make_tensor_env(env_name), max_episode_length, num_episodes, reduce_fn)
