"""Generation time of OpenES over MLP policies on a planar physics task, eager issue against graph replay.
usage: planar_es_timing.py [task] [pop] [steps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, workflows
from evox_amd.models import MLP
from evox_amd.problems.neuroevolution.reinforcement_learning import Brax, make_tensor_env

task = sys.argv[1] if len(sys.argv) > 1 else "hopper"
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda")
env = make_tensor_env(task)
net = MLP([env.obs_dim, 32, 32, env.act_dim])
for graph in (False, True):
    prob = Brax(net.batched_apply, task, T, 1)
    algo = algorithms.OpenES(net.init(ex.PRNGKey(1), device=dev), pop, 0.05, 0.1)
    wf = workflows.StdWorkflow(algo, prob, opt_direction="max", graph_step=graph)
    state = wf.init(ex.PRNGKey(2).to(dev))
    for _ in range(6):
        state = wf.step(state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        state = wf.step(state)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"task": task, "pop": pop, "episode_steps": T, "policy": [env.obs_dim, 32, 32, env.act_dim], "graph": graph,
                      "ms_per_generation": round(ms, 3), "env_steps_per_s": round(pop * T / ms * 1e3)}), flush=True)
