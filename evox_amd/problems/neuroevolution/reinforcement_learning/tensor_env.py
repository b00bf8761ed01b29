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
        for t in range(self.max_episode_length):
            action = self.policy(weights, obs.reshape(pop, E, -1)).reshape(pop * E, -1)
            env_state, obs, reward, step_done = self.env.step(env_state, action)
            total = total + torch.where(done, torch.zeros_like(reward), reward.to(torch.float32))
            done = done | step_done
            if self.early_exit_every and (t + 1) % self.early_exit_every == 0 and bool(done.all()):
                break
        total = self.reduce_fn(total.reshape(pop, E), dim=-1)
        if isinstance(total, tuple):         # torch.max / torch.median return (values, indices)
            total = total[0]
        return total, state.replace(key=key)


class Brax(TensorEnvProblem):
    """Name-based constructor with the reference's signature (``brax.py:16-24``).  ``env_name``
    is looked up in the tensor-environment registry (``register_tensor_env``); the real Brax
    engine needs JAX and is not available to this framework."""

    def __init__(self, policy, env_name, max_episode_length, num_episodes, reduce_fn=torch.mean, backend="generalized"):
        self.env_name, self.backend = env_name, backend
        super().__init__(policy, make_tensor_env(env_name), max_episode_length, num_episodes, reduce_fn)
