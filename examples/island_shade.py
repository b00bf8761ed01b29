"""Island model: one independent SHADE population per process, the best individuals migrate every ten generations.

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 examples/island_shade.py

Started as a plain script it is a single island.
"""
import os

import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems
from evox_amd.core import distributed as dist
from evox_amd.workflows import IslandWorkflow

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    dist.init_process_group()
rank, world = dist.get_process_id(), dist.get_world_size()
device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

dim = 20
monitor = monitors.EvalMonitor()
algorithm = algorithms.SHADE(torch.full((dim,), -5.0, device=device), torch.full((dim,), 5.0, device=device), pop_size=128)
workflow = IslandWorkflow(algorithm, problems.numerical.Rastrigin(), monitors=[monitor], migrate_every=10, num_migrants=4)
state = workflow.init(ex.PRNGKey(11))
for _ in range(200):
    state = workflow.step(state)
print(f"island {rank} of {world}: best fitness", float(monitor.get_best_fitness()))
