"""Host-side cost of one generation of any algorithm on the GPU: issue time vs wall time over 200
steps and a cProfile of 200 more.  usage: cpu_profile_algo.py <Algorithm> [pop] [dim]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows

name = sys.argv[1]
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
lb, ub = torch.full((dim,), -100.0, device=dev), torch.full((dim,), 100.0, device=dev)
try:
    algo = getattr(algorithms, name)(lb, ub, pop_size=pop)
except TypeError:
    algo = getattr(algorithms, name)(lb, ub, pop)
prob = problems.numerical.CEC2022TestSuit.create(1) if dim in (2, 10, 20) else problems.numerical.Rastrigin()
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(0).to(dev))
for _ in range(30):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    state = wf.step(state)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f"{name} pop={pop} dim={dim}: issue {t_issue * 5e3:.1f} us/step, wall {t_total * 5e3:.1f} us/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    state = wf.step(state)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:9000])
