"""NSGA-II on ZDT1 (two objectives) at large populations: time per generation, and the non-dominated sort of the merged
population alone.  usage: nsga2_m2_timing.py [pop ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows
from evox_amd.ops import nds

dev = torch.device("cuda")
for pop in [int(a) for a in sys.argv[1:]] or [10000, 50000, 100000]:
    d = 30
    algo = algorithms.NSGA2(lb=torch.zeros(d, device=dev), ub=torch.ones(d, device=dev), n_objs=2, pop_size=pop)
    wf = workflows.StdWorkflow(algo, problems.numerical.ZDT1(n=d))
    state = wf.init(ex.PRNGKey(1).to(dev))
    for _ in range(5):
        state = wf.step(state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        state = wf.step(state)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    g = torch.Generator().manual_seed(pop)
    x = torch.rand(2 * pop, 2, generator=g).to(dev)
    row = {"pop": pop, "nsga2_zdt1_ms_per_generation": round(ms, 3)}
    for label, until in (("full_rank_uniform_2pop_ms", None), ("rank_until_pop_uniform_2pop_ms", pop)):
        f = lambda: nds.non_dominated_rank(x, until)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        row[label] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    row["fronts_uniform"] = int(nds.non_dominated_rank(x).max()) + 1
    print(json.dumps(row), flush=True)
