"""Host-side profile (cProfile, cumulative) of a generation of a multi-objective algorithm on DTLZ2 (m = 3).
usage: mo_host_profile.py NAME [pop]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import evox_amd as ex
from evox_amd import algorithms, problems, workflows

name = sys.argv[1]
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
algo = algorithms.LMOCSO(3, lb, ub, pop) if name == "LMOCSO" else getattr(algorithms, name)(lb, ub, 3, pop)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(0).to(dev))
for _ in range(10):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    state = wf.step(state)
torch.cuda.synchronize()
print(f"== {name} pop {pop}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per generation")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    state = wf.step(state)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
