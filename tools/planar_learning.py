"""Does policy search on the planar physics tasks learn?  OpenES (rank-shaped fitness, mirrored sampling, Adam) over MLP
policies, graph-replayed generations on the GPU; one JSON line per task: the return of the search centre at the start
and after the run.  usage: planar_learning.py [generations] [task ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, monitors, workflows
from evox_amd.models import MLP
from evox_amd.problems.neuroevolution.reinforcement_learning import Brax, make_tensor_env
from evox_amd.utils import rank_based_fitness

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 150
tasks = sys.argv[2:] or ["inverted_pendulum", "inverted_double_pendulum", "reacher", "swimmer", "hopper", "walker2d", "halfcheetah", "ant"]
dev = torch.device("cuda")
for task in tasks:
    env = make_tensor_env(task)
    net = MLP([env.obs_dim, 32, 32, env.act_dim])
    T = 200
    prob = Brax(net.batched_apply, task, T, 1)
    w0 = net.init(ex.PRNGKey(1), device=dev)
    algo = algorithms.OpenES(w0, 1024, 0.03, 0.1, optimizer="adam", mirrored_sampling=True)
    wf = workflows.StdWorkflow(algo, prob, opt_direction="max", fitness_transforms=[rank_based_fitness], graph_step=True)
    state = wf.init(ex.PRNGKey(2).to(dev))

    def score(w):
        ev = Brax(net.batched_apply, task, T, 8)
        f, _ = ev.evaluate(ev.setup(ex.PRNGKey(99).to(dev)), w[None])
        return float(f[0])

    start = score(w0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    curve = []
    for g in range(gens):
        state = wf.step(state)
        if (g + 1) % max(gens // 5, 1) == 0:
            curve.append(round(score(state.get_child_state("algorithm").center), 2))
    torch.cuda.synchronize()
    print(json.dumps({"task": task, "generations": gens, "pop": 1024, "episode_steps": T, "return_at_start": round(start, 2),
                      "return_of_centre": curve, "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
