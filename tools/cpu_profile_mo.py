"""Host profile of one generation of a multi-objective algorithm on DTLZ2 (GPU)."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows

name = sys.argv[1]
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
algo = getattr(algorithms, name)(lb, ub, 3, pop)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(0).to(dev))
for _ in range(20):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    state = wf.step(state)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f"{name} pop={pop}: issue {t_issue * 1e4:.1f} us/step, wall {t_total * 1e4:.1f} us/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    state = wf.step(state)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:10000])
