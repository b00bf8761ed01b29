"""A few generations of one multi-objective algorithm on DTLZ2 (m=3), for profiling:  mo_step.py <algorithm> [pop] [gens]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import evox_amd as ex  # noqa: E402
from evox_amd import algorithms, metrics, problems, workflows  # noqa: E402

name = sys.argv[1]
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
algo = algorithms.LMOCSO(3, lb, ub, pop) if name == "LMOCSO" else getattr(algorithms, name)(lb, ub, 3, pop)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(0).to(dev))
for _ in range(3):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(gens):
    state = wf.step(state)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / gens * 1e3
fit = state.get_child_state("algorithm").fitness
fit = fit[torch.isfinite(fit).all(1)]
print(f"{name} pop={pop} ms/gen={ms:.3f} igd_after_{gens + 3}={float(metrics.igd(fit, prob.pf().to(dev))):.4f}", flush=True)
