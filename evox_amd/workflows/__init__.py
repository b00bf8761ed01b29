from .std_workflow import StdWorkflow, StdWorkflowState


def __getattr__(name):
    # the multi-process workflows import lazily (they pull in torch.distributed)
    if name in ("ShardedWorkflow", "IslandWorkflow"):
        from ..parallel import workflows as _w

        return getattr(_w, name)
    if name in ("RayDistributedWorkflow", "DistributedWorkflow"):
        from ..parallel import process_workflow as _p

        return getattr(_p, name)
    raise AttributeError(name)
