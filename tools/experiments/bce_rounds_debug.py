"""Rounds and time of the round-wise IBEA removal inside BCE-IBEA / IBEA generations."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import evox_amd as ex
from evox_amd import algorithms as A, problems, workflows
from evox_amd.ops import moselect as sel

name, pop = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda")
orig = sel.ibea_remove_rounds
log = []
def spy(obj, fit, C, kappa, n_remove, stats=None):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(obj, fit, C, kappa, n_remove, st)
    torch.cuda.synchronize()
    st["ms"] = round((time.perf_counter() - t0) * 1e3, 2); st["n"] = obj.shape[0]
    st["distinct"] = int(torch.unique(obj, dim=0).shape[0])
    log.append(st)
    return out
sel.ibea_remove_rounds = spy
prob = problems.numerical.DTLZ2(d=12, m=3)
algo = getattr(A, name)(torch.zeros(12, device=dev), torch.ones(12, device=dev), 3, pop)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(1).to(dev))
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 12
every = max(1, gens // 8)
for g in range(gens):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state = wf.step(state)
    torch.cuda.synchronize()
    print(g, "gen ms", round((time.perf_counter() - t0) * 1e3, 2), log[-1:] if g % every == 0 else "", flush=True)
    log.clear()
