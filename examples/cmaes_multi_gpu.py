"""CMA-ES with the population rows spread over all GPUs of a node.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/cmaes_multi_gpu.py

Each process owns one GPU, samples and evaluates its share of the rows and contributes its part of the
covariance statistics to one all-reduce per generation (RCCL).  Started as a plain script it runs as a
single process, on the GPU if there is one and on the CPU otherwise.
"""
import os

import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems
from evox_amd.core import distributed as dist
from evox_amd.workflows import ShardedWorkflow

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    dist.init_process_group()                     # nccl (= RCCL) when a GPU is visible, gloo otherwise
rank, world = dist.get_process_id(), dist.get_world_size()
device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

dim, pop_per_process = 50, 256
monitor = monitors.EvalMonitor()
algorithm = algorithms.CMAES(center_init=torch.full((dim,), 3.0, device=device), init_stdev=1.0,
                             pop_size=pop_per_process * world)
workflow = ShardedWorkflow(algorithm, problems.numerical.Sphere(), monitors=[monitor], resync_every=100)
state = workflow.init(ex.PRNGKey(7))              # the same key on every rank
for _ in range(200):
    state = workflow.step(state)
if rank == 0:
    print(f"{world} process(es), population {pop_per_process * world}: best fitness", float(monitor.get_best_fitness()))
