"""Policy-search problems.

* ``TensorEnvProblem`` – rollouts of device-resident batched environments (the role Brax plays
  for the reference); ``Brax`` resolves environment names through the same interface.
* ``planar_physics`` – a batched planar articulated-body engine and the seven planar Brax tasks on it
  (``inverted_pendulum``, ``inverted_double_pendulum``, ``reacher``, ``swimmer``, ``hopper``, ``walker2d``,
  ``halfcheetah``), registered under those names.
* ``spatial_physics`` – the same in space (free-floating root, tree of hinged bodies) and the ``ant`` task on it.
* ``Gym`` – arbitrary ``gymnasium``-API environments stepped by a pool of worker processes.
* ``EnvPool`` – a vectorised environment pool: the in-tree native C++ pool, or ``envpool`` when
  that package is installed."""
from .tensor_env import (Brax, SyntheticLocomotion, TensorEnv, TensorEnvProblem, make_tensor_env, register_tensor_env)
from .planar_physics import (PLANAR_TASKS, HalfCheetah, Hopper, InvertedDoublePendulum, InvertedPendulum, Link, PlanarEngine,
                             PlanarEnv, PlanarModel, Reacher, Swimmer, Walker2d)
from .spatial_physics import Ant, Body, SpatialEngine, SpatialModel
from .classic_envs import CartPoleEnv, PendulumEnv
from .gym import CapEpisode, Gym, Normalizer
from .env_pool import EnvPool, NativeEnvPool

__all__ = ["Brax", "SyntheticLocomotion", "TensorEnv", "TensorEnvProblem", "make_tensor_env", "register_tensor_env",
           "PLANAR_TASKS", "HalfCheetah", "Hopper", "InvertedDoublePendulum", "InvertedPendulum", "Link", "PlanarEngine",
           "PlanarEnv", "PlanarModel", "Reacher", "Swimmer", "Walker2d", "Ant", "Body", "SpatialEngine", "SpatialModel",
           "CartPoleEnv", "PendulumEnv", "CapEpisode", "Gym", "Normalizer", "EnvPool", "NativeEnvPool"]
