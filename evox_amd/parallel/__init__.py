from ..core.distributed import (all_gather, all_reduce, barrier, broadcast, get_process_id, get_world_size,
                                init_process_group, is_dist_initialized)
from .workflows import IslandWorkflow, ShardedWorkflow, supports_sharding
