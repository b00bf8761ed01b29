"""How often the CMA-ES recording is redone under graph_step, and what the polled decisions are."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import evox_amd as ex
from evox_amd import algorithms as A, problems, workflows

dev = torch.device("cuda")
dim, pop, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
graph = len(sys.argv) < 5 or sys.argv[4] != "eager"
algo = A.CMAES(torch.full((dim,), 3.0, device=dev), 1.0, pop_size=pop, decomposition="newton_schulz", precision=prec)
orig = algo.graph_retune
calls = []
def spy(state):
    out = orig(state)
    calls.append((int(state.ns_iters), int(state.ns_safe), int(state.ns_sym), [round(float(v), 4) for v in state.ns_res1.tolist()],
                  None if out is None else (int(out.ns_iters), int(out.ns_safe), int(out.ns_sym))))
    return out
algo.graph_retune = spy
wf = workflows.StdWorkflow(algo, problems.numerical.Rastrigin(), graph_step=graph)
state = wf.init(ex.PRNGKey(5).to(dev))
for _ in range(80):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    state = wf.step(state)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 5)
for c in calls:
    print(c)
