import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from evox_amd.ops import moselect as sel
from evox_amd.ops._util import hip, check, stream_of
dev = torch.device("cuda")
import evox_amd as ex
from evox_amd import algorithms as A, problems, workflows
# capture real knee_suppress inputs from a KnEA run
calls = []
orig = sel.knee_suppress
def spy(f, order, R):
    calls.append((f.clone(), order.clone(), R.clone()))
    return orig(f, order, R)
sel.knee_suppress = spy
from evox_amd.algorithms.mo import knea
knea._sel = sel if hasattr(knea, "_sel") else None
prob = problems.numerical.DTLZ2(d=12, m=3)
algo = A.KnEA(torch.zeros(12, device=dev), torch.ones(12, device=dev), 3, 10000)
wf = workflows.StdWorkflow(algo, prob)
state = wf.init(ex.PRNGKey(1).to(dev))
for g in range(30):
    state = wf.step(state)
print("calls", len(calls), "per gen", len(calls) / 30)
for f, order, R in calls[-6:]:
    n, m = f.shape
    fs = f[order].contiguous(); Rc = R.float().contiguous()
    st = torch.zeros(n, dtype=torch.uint8, device=dev); todo = torch.arange(n, device=dev)
    sizes = []
    while todo.numel() > 0:
        sizes.append(todo.numel())
        nxt = st.clone()
        check(hip().evox_hip_knee_state_round(fs.data_ptr(), Rc.data_ptr(), n, m, todo.data_ptr(), todo.numel(), st.data_ptr(), nxt.data_ptr(), 0, stream_of(fs)), "k")
        st = nxt; todo = torch.nonzero(st == 0).squeeze(1)
    for form in ("states", "rounds"):
        os.environ["EVOX_AMD_KNEE"] = form
        orig(f, order, R); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): orig(f, order, R)
        torch.cuda.synchronize(); print(form, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms", end="  ")
    print("n", n, "R", [round(float(v), 4) for v in R], "rounds", len(sizes), sizes[:12], "kept", int((st == 1).sum()))
