"""Which algorithms' steady-state step can be replayed from a HIP graph, and what it buys.

    python tools/graph_survey.py [--dim 100] [--pop 1024] [--steps 100] [--warmup 80]

For every algorithm: eager ms/step, graphed ms/step (pointer-stable ring replay, and the copy-in / clone-out
mode for comparison), and whether the graphed run reproduces the eager one exactly.  A step that synchronises with the host or keeps a changing Python value in
its state is reported with the reason the recording was refused."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex                                           # noqa: E402
from evox_amd import algorithms as A, problems, workflows     # noqa: E402
from evox_amd.core.tree import tree_leaves                    # noqa: E402


def make(name, dim, pop, dev):
    lb, ub = torch.full((dim,), -10.0, device=dev), torch.full((dim,), 10.0, device=dev)
    c = torch.full((dim,), 3.0, device=dev)
    table = {
        "PSO": lambda: A.PSO(lb, ub, pop), "CSO": lambda: A.CSO(lb, ub, pop), "CLPSO": lambda: A.CLPSO(lb, ub, pop, 0.5, 1.5, 0.05),
        "SLPSOGS": lambda: A.SLPSOGS(lb, ub, pop, 0.2, 0.7), "SLPSOUS": lambda: A.SLPSOUS(lb, ub, pop, 0.2, 0.7), "FIPS": lambda: A.FIPS(lb, ub, min(pop, 256)),
        "DE": lambda: A.DE(lb, ub, pop), "ODE": lambda: A.ODE(lb, ub, pop), "JaDE": lambda: A.JaDE(lb, ub, pop),
        "SHADE": lambda: A.SHADE(lb, ub, pop), "SaDE": lambda: A.SaDE(lb, ub, pop), "CoDE": lambda: A.CoDE(lb, ub, pop),
        "CMAES": lambda: A.CMAES(c, 1.0, pop_size=pop, decomposition="newton_schulz"), "SepCMAES": lambda: A.SepCMAES(c, 1.0, pop_size=pop),
        "OpenES": lambda: A.OpenES(c, pop, 0.05, 0.1), "PGPE": lambda: A.PGPE(pop, c, "adam", stdev_init=0.1),
        "SNES": lambda: A.SNES(pop, c), "XNES": lambda: A.XNES(c, torch.eye(dim, device=dev), pop_size=pop),
        "MAES": lambda: A.MAES(c, 1.0, pop_size=pop), "ARS": lambda: A.ARS(pop, c), "DES": lambda: A.DES(pop, c),
        "CR_FM_NES": lambda: A.CR_FM_NES(pop, c, 1.0),
        "DMSPSOEL": lambda: A.DMSPSOEL(lb, ub, pop // 16, 8, pop // 2, 5, 300, 0.7, 1.5, 1.5, 1.0, 1.0),
        "SwmmPSO": lambda: A.SwmmPSO(lb, ub, pop, shortcut=5), "FSPSO": lambda: A.FSPSO(lb, ub, pop),
        "IPOPCMAES": lambda: A.IPOPCMAES(c, 1.0, pop_size=pop), "BIPOPCMAES": lambda: A.BIPOPCMAES(c, 1.0, pop_size=pop),
        "LMMAES": lambda: A.LMMAES(c, 1.0, pop_size=pop), "AMaLGaM": lambda: A.AMaLGaM(c, 1.0, pop_size=pop),
        "IndependentAMaLGaM": lambda: A.IndependentAMaLGaM(c, 1.0, pop_size=pop), "RMES": lambda: A.RMES(c, 1.0, pop_size=pop),
        "ESMC": lambda: A.ESMC(pop + 1, c, lr=0.1, sigma=0.1), "SeparableNES": lambda: A.SeparableNES(c, torch.ones(dim, device=dev), pop),
        "GuidedES": lambda: A.GuidedES(pop, c, lrate_init=0.5, sigma_init=0.1),
        "Noise_reuse_es": lambda: A.Noise_reuse_es(pop, c, lr=0.1, sigma=0.1),
        "PersistentES": lambda: A.PersistentES(pop, c, lr=0.1, sigma=0.1, T=10, K=10),
        "ASEBO": lambda: A.ASEBO(pop, c, lr=0.05, sigma=0.1, subspace_dims=min(50, dim)),
        "LES": lambda: A.LES(pop, c, sigma_init=1.0, net_params=A.init_les_params(ex.PRNGKey(0).to(dev))),
    }
    return table[name]()


def run(name, args, dev, graph, mode="ring"):
    algo = make(name, args.dim, args.pop, dev)
    wf = workflows.StdWorkflow(algo, problems.numerical.Rastrigin(), graph_step=graph, graph_mode=mode)
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
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--pop", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=80)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    names = args.only.split(",") if args.only else ["PSO", "CSO", "CLPSO", "SLPSOGS", "SLPSOUS", "FIPS", "DE", "ODE", "JaDE", "SHADE",
                                                     "SaDE", "CoDE", "CMAES", "SepCMAES", "OpenES", "PGPE", "SNES", "XNES", "MAES",
                                                     "ARS", "DES", "CR_FM_NES", "DMSPSOEL", "SwmmPSO", "FSPSO", "IPOPCMAES",
                                                     "BIPOPCMAES", "LMMAES", "AMaLGaM", "IndependentAMaLGaM", "RMES", "ESMC",
                                                     "SeparableNES", "GuidedES", "Noise_reuse_es", "PersistentES", "ASEBO", "LES"]
    if not args.only:
        # one process per algorithm: a refused recording must not disturb the next measurement
        import subprocess
        for name in names:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--only", name, "--dim", str(args.dim), "--pop",
                            str(args.pop), "--steps", str(args.steps), "--warmup", str(args.warmup)], timeout=120)
        sys.exit(0)
    for name in names:
        row = {"algorithm": name}
        print(f"[graph_survey] {name}", flush=True)
        try:
            row["eager_ms"], se = run(name, args, dev, False)
        except Exception as e:                                   # noqa: BLE001
            row["eager_error"] = f"{type(e).__name__}: {e}"[:200]
            print(json.dumps(row), flush=True)
            continue
        try:
            row["graph_ms"], sg = run(name, args, dev, True)
            same = all(torch.equal(a, b) for a, b in zip(tree_leaves(se), tree_leaves(sg)) if isinstance(a, torch.Tensor))
            row["identical"] = bool(same)
            row["speedup"] = row["eager_ms"] / row["graph_ms"]
            row["graph_copy_mode_ms"], _ = run(name, args, dev, True, "copy")
        except Exception as e:                                   # noqa: BLE001
            row["graph_error"] = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.synchronize()
        print(json.dumps(row), flush=True)
