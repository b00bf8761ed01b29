"""Host-side cost of one headline generation: cProfile over 300 steps (GPU work is asynchronous,
so this is the launch/bookkeeping path that bounds the step once the kernels are fast)."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows

dev = torch.device("cuda")
center = torch.zeros(1000, device=dev) + 2.0
precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
algo = algorithms.CMAES(center_init=center, init_stdev=1.0, pop_size=10000, decomposition="newton_schulz", precision=precision)
wf = workflows.StdWorkflow(algo, problems.numerical.Ackley())
state = wf.init(ex.PRNGKey(0))
for _ in range(20):
    state = wf.step(state)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    state = wf.step(state)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f"back-to-back: loop time per step {t_issue*1e4:.1f} us, with the final synchronisation {t_total*1e4:.1f} us (100 steps; "
      "when the GPU is the bottleneck the launches block on the full queue and the two agree)")
# issue cost proper: every step is issued into an EMPTY queue (synchronise first), so no launch ever blocks
acc = 0.0
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = wf.step(state)
    acc += time.perf_counter() - t0
print(f"issue time per step into an empty queue {acc / 200 * 1e6:.1f} us (Python + ctypes + kernel launches, nothing waited for)")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    state = wf.step(state)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
