"""Worker of the multi-process tests: run by tests/test_distributed.py once per rank with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (gloo, CPU)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import evox_amd as ex
from evox_amd import algorithms as A
from evox_amd import monitors, problems, workflows
from evox_amd.core import distributed as dist
from evox_amd.parallel import IslandWorkflow, ShardedWorkflow


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_process_id(), dist.get_world_size()
    dim = 8
    # EVOX_TEST_DEVICE=cuda: every rank computes on the one visible GPU while the collectives stay
    # on gloo (device data is staged through host memory by evox_amd.core.distributed)
    dev = torch.device(os.environ.get("EVOX_TEST_DEVICE", "cpu"))
    c = torch.full((dim,), 3.0, device=dev)
    lb, ub = torch.full((dim,), -10.0, device=dev), torch.full((dim,), 10.0, device=dev)
    result = {}
    if case in ("cmaes", "openes", "pgpe", "pso_replicated"):
        algo = {"cmaes": lambda: A.CMAES(c, 1.0, pop_size=16, decomposition=os.environ.get("EVOX_TEST_DECOMP", "eigh")), "openes": lambda: A.OpenES(c, 16, 0.05, 0.1),
                "pgpe": lambda: A.PGPE(16, c, "adam"), "pso_replicated": lambda: A.PSO(lb, ub, 16)}[case]()
        mon = monitors.EvalMonitor()
        wf = ShardedWorkflow(algo, problems.numerical.Ackley(), monitors=[mon])
        state = wf.init(ex.PRNGKey(3))
        for _ in range(30):
            state = wf.step(state)
        alg = state.get_child_state("algorithm")
        vec = alg.mean if case == "cmaes" else (alg.global_best_location if case == "pso_replicated" else alg.center)
        result = {"vec": vec.tolist(), "world": state.world_size, "rank": state.rank,
                  "row_sharded": wf.row_sharded, "best": float(mon.get_best_fitness()),
                  "local_rows": int(alg.population.shape[0]) if case == "cmaes" else None}
    elif case == "std_multi_device":
        wf = workflows.StdWorkflow(A.PSO(lb, ub, 16), problems.numerical.Sphere())
        state = wf.enable_multi_devices(wf.init(ex.PRNGKey(5)))
        for _ in range(20):
            state = wf.step(state)
        alg = state.get_child_state("algorithm")
        result = {"vec": alg.global_best_location.tolist(), "world": state.world_size}
    elif case == "island":
        mon = monitors.EvalMonitor()
        wf = IslandWorkflow(A.SHADE(lb, ub, 32), problems.numerical.Sphere(), monitors=[mon], migrate_every=5,
                            num_migrants=4)
        state = wf.init(ex.PRNGKey(9))
        first = state.get_child_state("algorithm").population[0].tolist()
        for _ in range(40):
            state = wf.step(state)
        alg = state.get_child_state("algorithm")
        result = {"first": first, "best": float(alg.fitness.min())}
    elif case == "collectives":
        x = torch.arange(3, dtype=torch.float32) + 10 * rank
        g = dist.all_gather(x)
        s = dist.all_reduce(x)
        m = dist.all_reduce(x, "max")
        b = dist.broadcast(x, src=world - 1)
        g2 = dist.all_gather(x.reshape(1, 3), axis=1)
        result = {"gather": g.tolist(), "sum": s.tolist(), "max": m.tolist(), "bcast": b.tolist(),
                  "gather_axis1": list(g2.shape)}
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(result, f)
    dist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
