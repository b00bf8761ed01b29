from . import random, tree
from .algorithm import Algorithm, has_init_ask, has_init_tell
from .distributed import (POP_AXIS_NAME, ShardingType, all_gather, all_reduce, barrier, broadcast, get_process_id,
                          get_rank, get_world_size, init_process_group, is_dist_initialized, tree_all_gather,
                          tree_unpmap, unpmap)
from .module import Stateful, jit, jit_class, jit_method, use_state
from .monitor import Monitor
from .problem import Problem
from .state import State, get_state_sharding
from .struct import PyTreeNode, dataclass, pytree_field
from .tree import tree_flatten, tree_leaves, tree_map, tree_reduce, tree_structure, tree_unflatten
from .workflow import Workflow
