"""Two classic-control environments with the ``gymnasium`` single-environment API
(``reset(seed=) -> (obs, info)``, ``step(a) -> (obs, reward, terminated, truncated, info)``),
so that the ``Gym`` problem can be used and tested without the gymnasium package.  The
dynamics are the standard CartPole-v1 (Barto, Sutton & Anderson 1983) and Pendulum-v1
equations; the same equations run batched in the native pool (``csrc/envpool_cpu.cpp``)."""
from __future__ import annotations

import math

import numpy as np


class CartPoleEnv:
    obs_dim, act_dim, discrete = 4, 2, True
    max_steps = 500

    def __init__(self, max_steps=None):
        if max_steps is not None:
            self.max_steps = max_steps
        self.rng = np.random.default_rng(0)
        self.s = np.zeros(4, dtype=np.float64)
        self.t = 0

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.s = self.rng.uniform(-0.05, 0.05, size=4)
        self.t = 0
        return self.s.astype(np.float32), {}

    def step(self, action):
        x, x_dot, th, th_dot = self.s
        force = 10.0 if int(action) == 1 else -10.0
        g, mc, mp, l, tau = 9.8, 1.0, 0.1, 0.5, 0.02
        total, pml = mc + mp, mp * l
        c, s = math.cos(th), math.sin(th)
        temp = (force + pml * th_dot ** 2 * s) / total
        th_acc = (g * s - c * temp) / (l * (4.0 / 3.0 - mp * c * c / total))
        x_acc = temp - pml * th_acc * c / total
        self.s = np.array([x + tau * x_dot, x_dot + tau * x_acc, th + tau * th_dot, th_dot + tau * th_acc])
        self.t += 1
        terminated = bool(abs(self.s[0]) > 2.4 or abs(self.s[2]) > 12 * 2 * math.pi / 360)
        truncated = self.t >= self.max_steps
        return self.s.astype(np.float32), 1.0, terminated, truncated, {"x_position": float(self.s[0])}

    def render(self):
        return np.zeros((8, 8, 3), dtype=np.uint8)

    def close(self):
        pass


class PendulumEnv:
    obs_dim, act_dim, discrete = 3, 1, False
    max_steps = 200

    def __init__(self, max_steps=None):
        if max_steps is not None:
            self.max_steps = max_steps
        self.rng = np.random.default_rng(0)
        self.th, self.th_dot, self.t = 0.0, 0.0, 0

    def _obs(self):
        return np.array([math.cos(self.th), math.sin(self.th), self.th_dot], dtype=np.float32)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.th = self.rng.uniform(-math.pi, math.pi)
        self.th_dot = self.rng.uniform(-1.0, 1.0)
        self.t = 0
        return self._obs(), {}

    def step(self, action):
        u = float(np.clip(np.asarray(action).reshape(-1)[0], -2.0, 2.0))
        g, m, l, dt = 10.0, 1.0, 1.0, 0.05
        norm_th = ((self.th + math.pi) % (2 * math.pi)) - math.pi
        cost = norm_th ** 2 + 0.1 * self.th_dot ** 2 + 0.001 * u ** 2
        self.th_dot = float(np.clip(self.th_dot + (3 * g / (2 * l) * math.sin(self.th) + 3.0 / (m * l * l) * u) * dt, -8, 8))
        self.th = self.th + self.th_dot * dt
        self.t += 1
        return self._obs(), -cost, False, self.t >= self.max_steps, {"cost": cost}

    def render(self):
        return np.zeros((8, 8, 3), dtype=np.uint8)

    def close(self):
        pass
