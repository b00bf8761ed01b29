"""Which multi-objective algorithms' steady-state generation can be replayed from a HIP graph, and what it buys.

    python tools/graph_survey_mo.py [--pop 100] [--steps 100] [--warmup 20]

DTLZ2 with three objectives.  For every algorithm: eager ms per generation, graphed ms per generation (ring replay), and
whether the graphed run reproduces the eager one exactly; a generation that synchronises with the host is reported with
the reason the recording was refused.  One process per algorithm."""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex                                           # noqa: E402
from evox_amd import algorithms as A, problems, workflows     # noqa: E402
from evox_amd.core.tree import tree_leaves                    # noqa: E402

NAMES = ["NSGA2", "NSGA3", "RVEA", "RVEAa", "MOEAD", "MOEADDRA", "MOEADM2M", "EAGMOEAD", "IBEA", "BCEIBEA", "SPEA2", "HypE",
         "KnEA", "BiGE", "GDE3", "SRA", "TDEA", "LMOCSO", "IMMOEA"]


def run(name, args, dev, graph):
    prob = problems.numerical.DTLZ2(m=3)
    lb, ub = torch.zeros(prob.d, device=dev), torch.ones(prob.d, device=dev)
    algo = A.LMOCSO(3, lb, ub, args.pop) if name == "LMOCSO" else getattr(A, name)(lb, ub, 3, args.pop)
    wf = workflows.StdWorkflow(algo, prob, graph_step=graph, graph_mode="ring")
    state = wf.init(ex.PRNGKey(5).to(dev))
    for _ in range(args.warmup):
        state = wf.step(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = wf.step(state)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3, state


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=100)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if not args.only:
        for name in NAMES:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--only", name, "--pop", str(args.pop), "--steps",
                            str(args.steps), "--warmup", str(args.warmup)], timeout=180)
        sys.exit(0)
    dev = torch.device("cuda")
    name = args.only
    row = {"algorithm": name, "pop": args.pop}
    try:
        row["eager_ms"], se = run(name, args, dev, False)
        try:
            row["graph_ms"], sg = run(name, args, dev, True)
            def same(a, b):              # NaN rows (vectors without a solution) count as equal
                if a.is_floating_point():
                    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
                return torch.equal(a, b)
            row["identical"] = bool(all(same(a, b) for a, b in zip(tree_leaves(se), tree_leaves(sg))
                                        if isinstance(a, torch.Tensor)))
            row["speedup"] = row["eager_ms"] / row["graph_ms"]
        except Exception as e:                                   # noqa: BLE001
            row["graph_error"] = f"{type(e).__name__}: {e}"[:240]
    except Exception as e:                                       # noqa: BLE001
        row["eager_error"] = f"{type(e).__name__}: {e}"[:240]
    print(json.dumps(row), flush=True)
