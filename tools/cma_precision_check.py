"""Convergence of CMA-ES (pop 10^4, dim 10^3, Newton-Schulz factorisation) with fp32 and with bf16 GEMM operands.
usage: cma_precision_check.py [generations] ; one JSON line per (problem, precision): best fitness every 100 generations."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evox_amd as ex
from evox_amd import algorithms, problems, workflows

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda")
for pname in ("sphere", "ackley"):
    for precision in ("fp32", "bf16"):
        prob = problems.numerical.Sphere() if pname == "sphere" else problems.numerical.Ackley()
        g = torch.Generator().manual_seed(1)
        center = (torch.rand(1000, generator=g) * 4 - 2).to(dev)
        algo = algorithms.CMAES(center_init=center, init_stdev=1.0, pop_size=10000, precision=precision,
                                decomposition="newton_schulz")
        wf = workflows.StdWorkflow(algo, prob)
        state = wf.init(ex.PRNGKey(0).to(dev))
        trace = []
        for gen in range(1, gens + 1):
            state = wf.step(state)
            if gen % 100 == 0:
                a = state.get_child_state("algorithm")
                trace.append({"gen": gen, "f_mean": float(prob.evaluate(None, a.mean[None, :])[0][0]) if False else None,
                              "sigma": float(a.sigma)})
                fit, _ = prob.evaluate(state.get_child_state("problem"), a.mean[None, :])
                trace[-1]["f_mean"] = float(fit[0])
        print(json.dumps({"problem": pname, "precision": precision, "trace": trace}), flush=True)
