"""evox_amd — an evolutionary-computation framework built for AMD Instinct MI355X (gfx950).

Programming model: ``Algorithm.ask/tell`` · ``Problem.evaluate`` ·
``StdWorkflow.init(key)/step(state)`` over an immutable hierarchical ``State`` tree.
Tensors are ``torch.Tensor``; population-wide hot paths are hand-written HIP kernels
(``evox_amd.ops``), randomness is a counter-based Philox generator (``evox_amd.random``),
multi-GPU runs are one process per GPU over RCCL (``evox_amd.parallel``).
"""
from .core import (Algorithm, Monitor, Problem, PyTreeNode, ShardingType, State, Stateful, Workflow, dataclass,
                   get_state_sharding, has_init_ask, has_init_tell, jit, jit_class, jit_method, pytree_field,
                   random, tree, use_state)
from .core.random import PRNGKey

__version__ = "0.1.0"

from . import utils  # noqa: E402
from . import operators, problems, algorithms, workflows, monitors, metrics  # noqa: E402
