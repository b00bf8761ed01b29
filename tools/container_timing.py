"""Generation time of decomposition containers, eager and replayed from a HIP graph: the member loop (fuse=False) and,
where the base has one, the blocked / batched single instance (fuse="auto").
usage: container_timing.py [num_cluster] [dim] [pop]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms as A, problems, workflows

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda")
sub = dim // S
lb, ub = torch.full((sub,), -10.0, device=dev), torch.full((sub,), 10.0, device=dev)
for name, make in (("SHADE", lambda: A.SHADE(lb, ub, pop)), ("DE", lambda: A.DE(lb, ub, pop)),
                   ("SepCMAES", lambda: A.SepCMAES(torch.zeros(sub, device=dev), 1.0, pop_size=pop))):
  for fuse in (False, "auto"):
    row = {"base": name, "num_cluster": S, "dim": dim, "pop": pop, "fuse": fuse}
    for mode in (False, True):
        try:
            algo = A.ClusterdAlgorithm(make(), dim=dim, num_cluster=S, fuse=fuse)
            wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), graph_step=mode, graph_mode="ring")
            s = wf.init(ex.PRNGKey(1).to(dev))
            for _ in range(12):
                s = wf.step(s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                s = wf.step(s)
            torch.cuda.synchronize()
            row["graph_ms" if mode else "eager_ms"] = (time.perf_counter() - t0) / 30 * 1e3
            row["members"] = len(algo.members)
        except Exception as e:                                            # noqa: BLE001
            row["graph_error" if mode else "eager_error"] = f"{type(e).__name__}: {e}"[:160]
    # one member alone, for scale
    wf = workflows.StdWorkflow(make(), problems.numerical.Sphere())
    s = wf.init(ex.PRNGKey(1).to(dev))
    for _ in range(12):
        s = wf.step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        s = wf.step(s)
    torch.cuda.synchronize()
    row["one_member_eager_ms"] = (time.perf_counter() - t0) / 30 * 1e3
    print(json.dumps(row), flush=True)

# cooperative co-evolution: every sub-population is told a fitness of its own
for name, make in (("SHADE", lambda: A.SHADE(lb, ub, pop)), ("SepCMAES", lambda: A.SepCMAES(torch.zeros(sub, device=dev), 1.0, pop_size=pop))):
    for fuse in (False, "auto"):
        row = {"container": "VectorizedCoevolution", "base": name, "num_subpops": S, "dim": dim, "pop": pop, "fuse": fuse}
        for mode in (False, True):
            algo = A.VectorizedCoevolution(make(), dim=dim, num_subpops=S, fuse=fuse)
            wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), graph_step=mode, graph_mode="ring")
            s = wf.init(ex.PRNGKey(1).to(dev))
            for _ in range(12):
                s = wf.step(s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                s = wf.step(s)
            torch.cuda.synchronize()
            row["graph_ms" if mode else "eager_ms"] = (time.perf_counter() - t0) / 30 * 1e3
        print(json.dumps(row), flush=True)
