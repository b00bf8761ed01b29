"""Population and dimension scaling of whole generations on ONE GPU: how the time of a generation and the memory held
grow when the population or the decision vector grows by decades (CMA-ES, OpenES, PSO, CSO, NSGA-II).

    PYTHONPATH=. python3 tools/scaling_sweep.py [--out gpurun_out/scaling_single_gpu.jsonl] [--family cmaes,openes,pso,cso,nsga2]

One JSON line per case: ``ms_per_step``, ``evals_per_sec``, ``peak_mem_gb`` (torch allocator peak over the case) and
``bytes_per_eval`` = peak memory / population, the figure that says how many individuals fit next to the 288 GB."""
from __future__ import annotations

import argparse
import gc
import json
import sys
import time

import torch

import evox_amd as ex
from evox_amd import algorithms, problems, workflows


def timed(wf, state, budget=0.6, warmup=6, max_steps=400):
    for _ in range(warmup):
        state = wf.step(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = wf.step(state)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = int(max(3, min(max_steps, budget / max(one, 1e-6))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        state = wf.step(state)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, steps


def cases(dev):
    def cmaes(pop, dim):
        center = torch.full((dim,), 2.0, device=dev)
        algo = algorithms.CMAES(center_init=center, init_stdev=1.0, pop_size=pop, precision="bf16",
                                decomposition="newton_schulz")     # the headline benchmark's configuration
        return workflows.StdWorkflow(algo, problems.numerical.Ackley()), ex.PRNGKey(1).to(dev)

    def openes(pop, dim):
        algo = algorithms.OpenES(torch.zeros(dim, device=dev), pop_size=pop, learning_rate=0.01, noise_stdev=0.05,
                                 optimizer="adam", mirrored_sampling=True)
        return workflows.StdWorkflow(algo, problems.numerical.Sphere()), ex.PRNGKey(2).to(dev)

    def pso(pop, dim):
        algo = algorithms.PSO(lb=torch.full((dim,), -32.0, device=dev), ub=torch.full((dim,), 32.0, device=dev), pop_size=pop)
        return workflows.StdWorkflow(algo, problems.numerical.Ackley()), ex.PRNGKey(3).to(dev)

    def cso(pop, dim):
        algo = algorithms.CSO(lb=torch.full((dim,), -32.0, device=dev), ub=torch.full((dim,), 32.0, device=dev), pop_size=pop,
                              phi=0.1)
        return workflows.StdWorkflow(algo, problems.numerical.Ackley()), ex.PRNGKey(5).to(dev)

    def shade(pop, dim):
        algo = algorithms.SHADE(lb=torch.full((dim,), -32.0, device=dev), ub=torch.full((dim,), 32.0, device=dev), pop_size=pop)
        return workflows.StdWorkflow(algo, problems.numerical.Ackley()), ex.PRNGKey(6).to(dev)

    def de(pop, dim):
        algo = algorithms.DE(lb=torch.full((dim,), -32.0, device=dev), ub=torch.full((dim,), 32.0, device=dev), pop_size=pop,
                             batch_size=pop)
        return workflows.StdWorkflow(algo, problems.numerical.Ackley()), ex.PRNGKey(7).to(dev)

    def pgpe(pop, dim):
        algo = algorithms.PGPE(pop_size=pop, center_init=torch.zeros(dim, device=dev), optimizer="adam")
        return workflows.StdWorkflow(algo, problems.numerical.Sphere()), ex.PRNGKey(8).to(dev)

    def nsga2(pop, dim):
        algo = algorithms.NSGA2(lb=torch.zeros(dim, device=dev), ub=torch.ones(dim, device=dev), n_objs=3, pop_size=pop)
        return workflows.StdWorkflow(algo, problems.numerical.DTLZ2(m=3)), ex.PRNGKey(4).to(dev)

    grid = {
        "cmaes": (cmaes, [(1000, 1000), (10000, 1000), (100000, 1000),
                          (10000, 128), (10000, 512), (10000, 2048), (10000, 4096)]),
        "openes": (openes, [(1024, 10**5), (1024, 10**6), (1024, 10**7), (8192, 10**6), (65536, 10**5)]),
        "pso": (pso, [(10**4, 100), (10**5, 100), (10**6, 100), (10**7, 100), (10**5, 1000), (10**5, 10000)]),
        "cso": (cso, [(10**4, 100), (10**5, 100), (10**6, 100), (10**5, 1000), (10**4, 10**5)]),
        "shade": (shade, [(10**4, 100), (10**5, 100), (10**6, 100), (10**5, 1000)]),
        "de": (de, [(10**4, 100), (10**5, 100), (10**6, 100), (10**5, 1000)]),
        "pgpe": (pgpe, [(1024, 10**5), (1024, 10**6), (16384, 10**5)]),
        "nsga2": (nsga2, [(10**3, 12), (10**4, 12), (10**5, 12), (2 * 10**5, 12)]),
    }
    return grid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/scaling_single_gpu.jsonl")
    ap.add_argument("--family", default="cmaes,openes,pso,cso,nsga2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    grid = cases(dev)
    with open(args.out, "a") as out:
        for fam in args.family.split(","):
            make, shapes = grid[fam]
            for pop, dim in shapes:
                print(f"[scaling] {fam} pop={pop} dim={dim}", flush=True)
                gc.collect()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats()
                rec = {"family": fam, "pop": pop, "dim": dim}
                try:
                    wf, key = make(pop, dim)
                    state = wf.init(key)
                    dt, steps = timed(wf, state)
                    peak = torch.cuda.max_memory_allocated()
                    rec.update(ms_per_step=round(1e3 * dt, 4), steps=steps, evals_per_sec=round(pop / dt, 1),
                               peak_mem_gb=round(peak / 2**30, 3), bytes_per_eval=round(peak / pop, 1),
                               bytes_per_variable=round(peak / (pop * dim), 2))
                    del wf, state
                except Exception as e:                              # a refused shape is part of the answer
                    rec["error"] = f"{type(e).__name__}: {e}"[:300]
                out.write(json.dumps(rec) + "\n")
                out.flush()
                print(json.dumps(rec), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
