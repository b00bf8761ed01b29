"""Policy evaluation on ``gymnasium``-API environments stepped by worker processes.
Reference: ``src/evox/problems/neuroevolution/reinforcement_learning/gym.py:19-426`` (Ray actors).

Here the workers are plain OS processes (``multiprocessing`` *spawn* context, so that no worker
inherits an initialised GPU runtime) connected by pipes; the environment factory and the policy
travel as ``cloudpickle`` payloads.  Two modes, as in the reference:

* ``batch_policy=True``  – the controller (this process, typically on the GPU) evaluates the
  policy for all environments at once each step; workers only step environments.
* ``batch_policy=False`` – every worker receives its slice of the population and runs complete
  rollouts with the policy on its own CPU.

``num_workers=0`` steps the environments inside the calling process."""
from __future__ import annotations

import multiprocessing as mp
from typing import Callable, List, Optional

import numpy as np
import torch

from ....core import random as rnd
from ....core.module import Stateful, use_state
from ....core.problem import Problem
from ....core.state import State
from ....core.tree import tree_leaves, tree_map


def tree_batch_size(tree):
    return tree_leaves(tree)[0].shape[0]


class Normalizer(Stateful):
    """Running observation normaliser (reference ``gym.py:25-63``)."""

    def setup(self, key):
        return State(sum=0.0, sumOfSquares=0.0, count=0, mean=0.0)

    def _stats(self, state):
        mean = state.sum / state.count
        var = state.sumOfSquares / state.count - mean ** 2
        std = torch.sqrt(torch.clamp(torch.as_tensor(var), min=1e-2))
        return mean, std

    def normalize(self, state, x):
        state = state.replace(count=state.count + 1, sum=state.sum + x, sumOfSquares=state.sumOfSquares + x ** 2)
        mean, std = self._stats(state)
        return state.replace(mean=mean), (x - mean) / std

    def normalize_obvs(self, state, obvs):
        state = state.replace(count=state.count + len(obvs), sum=state.sum + obvs.sum(0),
                              sumOfSquares=state.sumOfSquares + (obvs ** 2).sum(0))
        mean, std = self._stats(state)
        return state.replace(mean=mean), (obvs - mean) / std


class CapEpisode(Stateful):
    """Adaptive episode cap: twice the mean episode length of the previous evaluation."""

    def __init__(self, init_cap=100):
        super().__init__()
        self.init_cap = init_cap

    def setup(self, key):
        return State(cap=self.init_cap)

    def update(self, state, episode_length):
        return state.replace(cap=int(round(float(torch.as_tensor(episode_length, dtype=torch.float32).mean()) * 2)))

    def get(self, state):
        return state.cap, state


class _EnvSet:
    """The environments owned by one worker (or by the calling process)."""

    def __init__(self, env_creator, policy, mo_keys):
        self.env_creator, self.policy, self.mo_keys = env_creator, policy, list(mo_keys)
        self.envs = []

    def reset(self, seed, num_env):
        while len(self.envs) < num_env:
            self.envs.append(self.env_creator())
        self.n = num_env
        self.total_rewards = np.zeros(num_env)
        self.acc_mo = np.zeros((num_env, len(self.mo_keys)))
        self.episode_length = np.zeros(num_env)
        self.terminated = np.zeros(num_env, dtype=bool)
        self.truncated = np.zeros(num_env, dtype=bool)
        self.observations = [env.reset(seed=seed)[0] for env in self.envs[:num_env]]
        return np.stack(self.observations)

    def step(self, actions):
        for i in range(self.n):
            if self.terminated[i] or self.truncated[i]:
                continue
            obs, reward, self.terminated[i], self.truncated[i], info = self.envs[i].step(actions[i])
            self.observations[i] = obs
            self.total_rewards[i] += reward
            self.episode_length[i] += 1
            for k, name in enumerate(self.mo_keys):
                if name not in info:
                    raise KeyError(f"mo_keys has a key {name}, which doesn't exist in `info`, "
                                   f"available fields are {list(info.keys())}.")
                self.acc_mo[i, k] += info[name]
        return np.stack(self.observations), self.terminated.copy(), self.truncated.copy()

    def results(self):
        return self.total_rewards, self.acc_mo, self.episode_length

    def rollout(self, seed, subpop, cap):
        n = tree_batch_size(subpop)
        obs = self.reset(seed, n)
        i = 0
        while True:
            with torch.no_grad():
                actions = self.policy(subpop, torch.as_tensor(obs))
            obs, term, trunc = self.step(np.asarray(actions))
            i += 1
            if np.all(term | trunc) or (cap and i >= cap):
                break
        return self.results()


def _worker_main(conn, payload):
    import cloudpickle
    torch.set_num_threads(1)
    envs = _EnvSet(*cloudpickle.loads(payload))
    while True:
        try:
            cmd, args = conn.recv()
        except EOFError:
            return
        if cmd == "close":
            conn.close()
            return
        try:
            conn.send(("ok", getattr(envs, cmd)(*args)))
        except Exception as e:                                  # surfaced in the controller
            conn.send(("err", repr(e)))


class _Remote:
    def __init__(self, ctx, payload):
        self.conn, child = ctx.Pipe()
        self.proc = ctx.Process(target=_worker_main, args=(child, payload), daemon=True)
        self.proc.start()
        child.close()

    def call_async(self, cmd, *args):
        self.conn.send((cmd, args))

    def result(self):
        status, value = self.conn.recv()
        if status == "err":
            raise RuntimeError(f"gym worker failed: {value}")
        return value

    def close(self):
        try:
            self.conn.send(("close", ()))
        except (BrokenPipeError, OSError):
            pass
        self.proc.join(timeout=5)
        if self.proc.is_alive():
            self.proc.kill()


class _Local:
    def __init__(self, envs):
        self.envs, self._out = envs, None

    def call_async(self, cmd, *args):
        self._out = getattr(self.envs, cmd)(*args)

    def result(self):
        return self._out

    def close(self):
        pass


def _split_sizes(n, parts):
    base, rem = divmod(n, parts)
    return [base + 1 if i < rem else base for i in range(parts)]


class Gym(Problem):
    def __init__(self, policy: Callable, num_workers: int, env_name: Optional[str] = None, env_options: dict = {},
                 env_creator: Optional[Callable] = None, mo_keys: List = [], controller_options: dict = {},
                 worker_options: dict = {}, init_cap: Optional[int] = None, batch_policy: bool = False):
        super().__init__()
        if env_name:
            def env_creator(env_name=env_name, env_options=dict(env_options)):
                import gymnasium
                return gymnasium.make(env_name, **env_options)
        if not env_creator:
            raise ValueError("Either 'env_name' or 'env_creator' must be set.'")
        self.policy, self.mo_keys, self.batch_policy = policy, list(mo_keys), batch_policy
        self.env_name, self.env_creator = env_name, env_creator
        self.num_workers = num_workers
        if num_workers <= 0:
            self.workers = [_Local(_EnvSet(env_creator, policy, mo_keys))]
        else:
            import cloudpickle
            payload = cloudpickle.dumps((env_creator, None if batch_policy else policy, mo_keys))
            ctx = mp.get_context("spawn")
            self.workers = [_Remote(ctx, payload) for _ in range(num_workers)]
        self.cap_episode = CapEpisode(init_cap=init_cap) if init_cap is not None else None

    def setup(self, key):
        return State(key=key)

    # ------------------------------------------------------------------------------------
    def _gather(self):
        return [w.result() for w in self.workers]

    def _batched_rollouts(self, seed, pop, cap):
        pop_size = tree_batch_size(pop)
        device = tree_leaves(pop)[0].device
        sizes = _split_sizes(pop_size, len(self.workers))
        for w, n in zip(self.workers, sizes):
            w.call_async("reset", seed, n)
        obs = np.concatenate([o for o, n in zip(self._gather(), sizes) if n > 0], axis=0)
        bounds = np.cumsum([0] + sizes)
        i = 0
        while True:
            with torch.no_grad():
                actions = self.policy(pop, torch.as_tensor(obs).to(device)).cpu().numpy()
            for w, a, b in zip(self.workers, bounds[:-1], bounds[1:]):
                w.call_async("step", actions[a:b])
            outs = self._gather()
            obs = np.concatenate([o[0] for o, n in zip(outs, sizes) if n > 0], axis=0)
            finished = np.concatenate([o[1] | o[2] for o in outs])
            i += 1
            if finished.all() or (cap and i >= cap):
                break
        for w in self.workers:
            w.call_async("results")
        return self._gather()

    def _independent_rollouts(self, seed, pop, cap):
        pop_size = tree_batch_size(pop)
        bounds = np.cumsum([0] + _split_sizes(pop_size, len(self.workers)))
        used = []
        for w, a, b in zip(self.workers, bounds[:-1], bounds[1:]):
            if b > a:
                sub = tree_map(lambda t: t[a:b].detach().cpu(), pop)
                w.call_async("rollout", seed, sub, cap)
                used.append(w)
        return [w.result() for w in used]

    def evaluate(self, state, pop):
        key, subkey = rnd.split(state.key)
        seed = int(rnd.randint(subkey, (1,), 0, 2 ** 31 - 1)[0])
        cap = None
        if self.cap_episode is not None:
            cap, state = use_state(self.cap_episode.get)(state)
        run = self._batched_rollouts if self.batch_policy else self._independent_rollouts
        parts = run(seed, pop, cap)
        rewards = np.concatenate([p[0] for p in parts])
        acc_mo = np.concatenate([p[1] for p in parts], axis=0)
        lengths = np.concatenate([p[2] for p in parts])
        device = tree_leaves(pop)[0].device
        if self.cap_episode is not None:
            state = use_state(self.cap_episode.update)(state, lengths)
        out = acc_mo if self.mo_keys else rewards
        return torch.as_tensor(out, dtype=torch.float32).to(device), state.replace(key=key)

    def visualize(self, key, weights, ale_render_mode="rgb_array"):
        """Run one episode with a single weight set and return the rendered frames."""
        seed = int(rnd.randint(key, (1,), 0, 2 ** 31 - 1)[0])
        env = self.env_creator()
        obs, _ = env.reset(seed=seed)
        frames = [env.render()] if ale_render_mode == "rgb_array" else []
        done = False
        batched = tree_map(lambda t: t[None], weights)
        while not done:
            action = np.asarray(self.policy(batched, torch.as_tensor(obs)[None]))[0]
            obs, _r, terminated, truncated, _ = env.step(action)
            done = terminated or truncated
            if ale_render_mode == "rgb_array":
                frames.append(env.render())
        return frames

    def close(self):
        for w in self.workers:
            w.close()
        self.workers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
