"""NSGA-II on 3-objective DTLZ2 with IGD tracking."""
import torch

import evox_amd as ex
from evox_amd import algorithms, metrics, problems, workflows

device = "cuda" if torch.cuda.is_available() else "cpu"
problem = problems.numerical.DTLZ2(m=3)
algorithm = algorithms.NSGA2(lb=torch.zeros(12, device=device), ub=torch.ones(12, device=device), n_objs=3, pop_size=1000)
workflow = workflows.StdWorkflow(algorithm, problem)
state = workflow.init(ex.PRNGKey(0))
igd = metrics.IGD(problem.pf().to(device))
for gen in range(1, 101):
    state = workflow.step(state)
    if gen % 20 == 0:
        print(gen, float(igd(state.get_child_state("algorithm").fitness)))
