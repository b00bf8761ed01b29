"""OpenES training an MLP policy on the planar hopper (in-tree articulated-body physics, device-resident rollouts; set
TASK=ant, walker2d, halfcheetah, swimmer, ... for another task); the whole generation is replayed from a HIP graph on a
GPU, and the best policy's episode is written as an HTML page."""
import os

import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, workflows
from evox_amd.models import MLP
from evox_amd.problems.neuroevolution import Brax
from evox_amd.utils import rank_based_fitness

device = "cuda" if torch.cuda.is_available() else "cpu"
from evox_amd.problems.neuroevolution import make_tensor_env

task = os.environ.get("TASK", "hopper")
env = make_tensor_env(task)
net = MLP([env.obs_dim, 32, 32, env.act_dim])
problem = Brax(policy=net.batched_apply, env_name=task, max_episode_length=200, num_episodes=1)
algorithm = algorithms.OpenES(net.init(ex.PRNGKey(1)).to(device), pop_size=1024, learning_rate=0.05, noise_stdev=0.1,
                              optimizer="adam", mirrored_sampling=True)
monitor = monitors.EvalMonitor()
workflow = workflows.StdWorkflow(algorithm, problem, monitors=[monitor], opt_direction="max",
                                 fitness_transforms=[rank_based_fitness], graph_step=device == "cuda")
state = workflow.init(ex.PRNGKey(0).to(device))
for gen in range(1, int(os.environ.get("GENERATIONS", "100")) + 1):
    state = workflow.step(state)
    if gen % 10 == 0:
        print(gen, "best return so far", float(monitor.get_best_fitness()))
best = state.get_child_state("algorithm").center
with open(f"{task}_episode.html", "w") as f:
    f.write(problem.visualize(ex.PRNGKey(2).to(device), best, output_type="HTML"))
print(f"wrote {task}_episode.html")
