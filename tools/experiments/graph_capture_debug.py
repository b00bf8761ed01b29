import sys, traceback, torch
sys.path.insert(0, ".")
import evox_amd as ex
from evox_amd import algorithms as A, problems, workflows
name = sys.argv[1]
dev = torch.device("cuda")
prob = problems.numerical.DTLZ2(m=3)
lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
algo = A.LMOCSO(3, lb, ub, 100) if name == "LMOCSO" else getattr(A, name)(lb, ub, 3, 100)
wf = workflows.StdWorkflow(algo, prob, graph_step=True)
state = wf.init(ex.PRNGKey(5).to(dev))
try:
    for _ in range(6):
        state = wf.step(state)
    print(name, "OK")
except Exception as e:
    tb = e.__cause__ or e.__context__ or e
    lines = traceback.format_exception(type(tb), tb, tb.__traceback__)
    keep = [l for l in lines if "evox_amd" in l]
    print(name, "FAIL", type(tb).__name__, "".join(keep[-3:]))
