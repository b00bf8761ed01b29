"""Vectorised-environment policy evaluation.
Reference: ``src/evox/problems/neuroevolution/reinforcement_learning/env_pool.py:14-80`` (wraps the
external ``envpool`` C++ library).  This framework carries its own native pool
(``_native/csrc/envpool_cpu.cpp``: CartPole-v1, Pendulum-v1; OpenMP over environments) and falls
back to ``envpool`` for other names when that package is importable.  The policy runs batched on
the population's device; observations/actions cross the host boundary once per step."""
from __future__ import annotations

import ctypes
from typing import Callable, Optional

import numpy as np
import torch

from .... import _native
from ....core import random as rnd
from ....core.problem import Problem
from ....core.state import State
from ....core.tree import tree_leaves

P, I, I64 = _native.P, _native.I, _native.I64
_native.register_cpu_api({
    "evox_cpu_envpool_create": (P, [I, I, I]),
    "evox_cpu_envpool_destroy": (None, [P]),
    "evox_cpu_envpool_obs_dim": (I, [P]),
    "evox_cpu_envpool_act_dim": (I, [P]),
    "evox_cpu_envpool_seed": (None, [P, ctypes.c_uint64]),
    "evox_cpu_envpool_reset": (None, [P, P]),
    "evox_cpu_envpool_step": (None, [P, P, P, P, P, P]),
})

_NATIVE_IDS = {"CartPole-v1": (0, 500), "CartPole-v0": (0, 200), "Pendulum-v1": (1, 200)}


class NativeEnvPool:
    """``gymnasium``-vector style API over the native pool."""

    def __init__(self, env_name: str, num_envs: int, max_episode_steps: Optional[int] = None):
        if env_name not in _NATIVE_IDS:
            raise KeyError(f"native pool has {sorted(_NATIVE_IDS)}, got {env_name!r}")
        env_id, default_steps = _NATIVE_IDS[env_name]
        self.lib = _native.cpu()
        self.num_envs = num_envs
        self.handle = self.lib.evox_cpu_envpool_create(env_id, num_envs, max_episode_steps or default_steps)
        if not self.handle:
            raise RuntimeError("could not create native environment pool")
        self.obs_dim = self.lib.evox_cpu_envpool_obs_dim(self.handle)
        self.discrete = env_id == 0
        self.obs = np.empty((num_envs, self.obs_dim), dtype=np.float32)
        self.reward = np.empty(num_envs, dtype=np.float32)
        self.terminated = np.zeros(num_envs, dtype=np.uint8)
        self.truncated = np.zeros(num_envs, dtype=np.uint8)

    def seed(self, seed):
        self.lib.evox_cpu_envpool_seed(self.handle, int(np.asarray(seed).reshape(-1)[0]))

    def reset(self, env_ids=None):
        self.terminated[:] = 0
        self.truncated[:] = 0
        self.lib.evox_cpu_envpool_reset(self.handle, self.obs.ctypes.data)
        return self.obs.copy(), {}

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1)[:, 0])
        self.lib.evox_cpu_envpool_step(self.handle, a.ctypes.data, self.obs.ctypes.data, self.reward.ctypes.data,
                                       self.terminated.ctypes.data, self.truncated.ctypes.data)
        return (self.obs.copy(), self.reward.copy(), self.terminated.astype(bool), self.truncated.astype(bool), {})

    def close(self):
        if getattr(self, "handle", None):
            self.lib.evox_cpu_envpool_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def _make_pool(env_name, num_envs, env_options):
    if env_name in _NATIVE_IDS:
        return NativeEnvPool(env_name, num_envs, **env_options)
    try:
        import envpool
    except ImportError as e:
        raise ImportError(f"{env_name!r} is not in the native pool ({sorted(_NATIVE_IDS)}) and envpool is not "
                          f"installed: {e}")
    return envpool.make(env_name, num_envs=num_envs, env_type="gymnasium", **env_options)


class EnvPool(Problem):
    """``policy(pop, obs)`` is batched: individual i acts in environment i (``num_envs`` must
    equal the population size)."""

    def __init__(self, policy: Callable, env_name: str, num_envs: int, env_options: dict = {},
                 cap_episode_length: Optional[int] = None):
        super().__init__()
        self.policy = policy
        self.num_envs = num_envs
        self.env = _make_pool(env_name, num_envs, dict(env_options))
        self.cap_episode_length = cap_episode_length

    def setup(self, key):
        return State(key=key)

    def evaluate(self, state, pop):
        key, subkey = rnd.split(state.key)
        device = tree_leaves(pop)[0].device
        self.env.seed(int(rnd.randint(subkey, (1,), 0, 2 ** 31 - 1)[0]))
        obs, _ = self.env.reset(None)
        total = np.zeros(self.num_envs, dtype=np.float64)
        done = np.zeros(self.num_envs, dtype=bool)
        i = 0
        while not done.all() and not (self.cap_episode_length and i >= self.cap_episode_length):
            with torch.no_grad():
                action = self.policy(pop, torch.as_tensor(obs).to(device)).cpu().numpy()
            obs, reward, terminated, truncated, _ = self.env.step(action)
            total += np.where(done, 0.0, reward)
            done |= terminated | truncated
            i += 1
        return torch.as_tensor(total, dtype=torch.float32).to(device), state.replace(key=key)
