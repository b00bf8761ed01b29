"""OpenES training an MLP policy on the in-tree synthetic locomotion task (device-resident rollouts)."""
import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, workflows
from evox_amd.models import MLP
from evox_amd.problems.neuroevolution import Brax
from evox_amd.utils import rank_based_fitness

device = "cuda" if torch.cuda.is_available() else "cpu"
net = MLP([27, 64, 64, 8])
problem = Brax(policy=net.batched_apply, env_name="synthetic-ant", max_episode_length=100, num_episodes=2)
algorithm = algorithms.OpenES(net.init(ex.PRNGKey(1)).to(device), pop_size=256, learning_rate=0.05, noise_stdev=0.05,
                              optimizer="adam", mirrored_sampling=True)
monitor = monitors.EvalMonitor()
workflow = workflows.StdWorkflow(algorithm, problem, monitors=[monitor], opt_direction="max",
                                 fitness_transforms=[rank_based_fitness])
state = workflow.init(ex.PRNGKey(0))
for gen in range(1, 51):
    state = workflow.step(state)
    if gen % 10 == 0:
        print(gen, "mean return", float(monitor.get_latest_fitness().mean()))
