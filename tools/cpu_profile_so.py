"""Host-side cost of one generation of a single-objective algorithm on the GPU (constructors of
``graph_survey.make``): issue time vs wall time over 100 steps and a cProfile of 100 more.
usage: cpu_profile_so.py <Algorithm> [pop] [dim]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import problems, workflows
from graph_survey import make

name = sys.argv[1]
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda")
wf = workflows.StdWorkflow(make(name, dim, pop, dev), problems.numerical.Rastrigin())
state = wf.init(ex.PRNGKey(5).to(dev))
for _ in range(20):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    state = wf.step(state)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f"{name} pop={pop} dim={dim}: issue {t_issue * 1e4:.1f} us/step, wall {t_total * 1e4:.1f} us/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    state = wf.step(state)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:5000])
