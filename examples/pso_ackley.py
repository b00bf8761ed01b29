"""PSO on Ackley (the reference's README example) – runs on CPU or GPU."""
import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems, workflows

device = "cuda" if torch.cuda.is_available() else "cpu"
monitor = monitors.EvalMonitor()
workflow = workflows.StdWorkflow(
    algorithms.PSO(lb=torch.full((2,), -32.0, device=device), ub=torch.full((2,), 32.0, device=device), pop_size=100),
    problems.numerical.Ackley(), monitors=[monitor])
state = workflow.init(ex.PRNGKey(42))
for _ in range(100):
    state = workflow.step(state)
print("best fitness:", float(monitor.get_best_fitness()))
