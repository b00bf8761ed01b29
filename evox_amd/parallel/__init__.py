from ..core.distributed import (all_gather, all_reduce, barrier, broadcast, get_process_id, get_world_size,
                                init_process_group, is_dist_initialized)
from .workflows import IslandWorkflow, ShardedWorkflow, supports_sharding


def __getattr__(name):
    # the peer-to-peer group needs the HIP library; resolve it on first use
    if name in ("P2PGroup", "P2PTimeout"):
        from . import p2p

        return getattr(p2p, name)
    raise AttributeError(name)
