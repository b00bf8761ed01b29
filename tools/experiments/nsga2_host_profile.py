"""Host-side profile of an NSGA-II generation at population 10^5 (where does the wall time go beyond the kernels)."""
import cProfile
import pstats
import sys

import torch

import evox_amd as ex
from evox_amd import algorithms, problems, workflows

pop = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
algo = algorithms.NSGA2(torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev), 3, pop)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(0).to(dev))
for _ in range(10):
    state = wf.step(state)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    state = wf.step(state)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
