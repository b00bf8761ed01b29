"""Decomposition container: a 200-dimensional problem split into 20 blocks of 10 variables, one SHADE per block.

The 20 members run as ONE blocked instance (``algorithm.fused``): every block draws its own parents, crossover masks and
control parameters and keeps its own success-history memory, inside the launches of a single member.
``fuse=False`` runs the members one after the other instead (the same search, 20 times the launches).
"""
import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems, workflows

device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
block = 10
member = algorithms.SHADE(torch.full((block,), -32.0, device=device), torch.full((block,), 32.0, device=device), pop_size=100)
algorithm = algorithms.ClusterdAlgorithm(member, dim=200, num_cluster=20)
monitor = monitors.EvalMonitor()
# a generation is a dozen short launches: replayed from a HIP graph on the GPU (ignored on the CPU)
workflow = workflows.StdWorkflow(algorithm, problems.numerical.Ackley(), monitors=[monitor], graph_step="auto")
state = workflow.init(ex.PRNGKey(5).to(device))
for _ in range(300):
    state = workflow.step(state)
print(f"members fused into one instance: {algorithm.fused}; best fitness", float(monitor.get_best_fitness()))
