"""Minimal torque-limited pendulum with the gym 0.17 `Env` calling convention the reference's
run_env expects (reset() -> obs, step(a) -> (obs, reward, done, info), spaces with .low/.high).
gym is not available here; this is a fresh implementation of the textbook dynamics used for the
closed-loop smoke test, not a copy of gym's Pendulum-v0 source."""
from types import SimpleNamespace

import numpy as np


class PendulumEnv:
    max_speed = 8.0
    max_torque = 2.0
    dt = 0.05
    g = 10.0
    m = 1.0
    length = 1.0

    def __init__(self, seed=0):
        hi = np.array([1.0, 1.0, self.max_speed])
        self.observation_space = SimpleNamespace(low=-hi, high=hi, shape=(3,))
        self.action_space = SimpleNamespace(low=np.array([-self.max_torque]), high=np.array([self.max_torque]), shape=(1,))
        self.rng = np.random.default_rng(seed)
        self.state = None

    def _obs(self):
        th, thdot = self.state
        return np.array([np.cos(th), np.sin(th), thdot])

    def reset(self):
        self.state = np.array([self.rng.uniform(-np.pi, np.pi), self.rng.uniform(-1, 1)])
        return self._obs()

    def step(self, action):
        th, thdot = self.state
        u = float(np.clip(np.asarray(action).reshape(-1)[0], -self.max_torque, self.max_torque))
        ang = ((th + np.pi) % (2 * np.pi)) - np.pi
        cost = ang ** 2 + 0.1 * thdot ** 2 + 0.001 * u ** 2
        thdot = thdot + (3 * self.g / (2 * self.length) * np.sin(th) + 3.0 / (self.m * self.length ** 2) * u) * self.dt
        thdot = float(np.clip(thdot, -self.max_speed, self.max_speed))
        self.state = np.array([th + thdot * self.dt, thdot])
        return self._obs(), -cost, False, {}

    def __exit__(self, *a):
        return False
