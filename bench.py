"""Headline benchmark: CMA-ES, pop = 10 000 per GPU, dim = 1 000, Ackley — evaluations/s.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU over RCCL.  A "step" is one full generation (sample λ candidates,
evaluate them, sort, update mean / paths / covariance / step size, factorise C) with the
reference's default hyper-parameters (μ = λ/2, factorisation every generation for this
size).  The factorisation yields C^{1/2} and C^{-1/2}, the two matrices the update equations
use: by default through a Newton–Schulz iteration (its residual is reported), with
``--decomp eigh`` through a full eigendecomposition as the reference does.  Synthetic objective (Ackley) — no data set is involved; the distribution starts from a
random mean.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _self_launch(n: int) -> int:
    """Run this script as ``n`` ranks on this node (``evox_amd/utils/launch.py``, loaded by path so that this
    parent process imports neither torch nor the package)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "_evox_launch", os.path.join(os.path.dirname(os.path.abspath(__file__)), "evox_amd", "utils", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.launch_ranks(__file__, sys.argv[1:], n)


def _secondary(dev):
    """The other benchmark configurations on the same GPU, timed like the headline (synchronise, K generations,
    synchronise).  Each entry is measured on its own; a failure is recorded in place of its numbers."""
    import torch

    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows

    def timed(wf, state, steps, warmup):
        for _ in range(warmup):
            state = wf.step(state)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            state = wf.step(state)
        torch.cuda.synchronize()
        return state, time.perf_counter() - t0

    def nsga2():
        pop, d, steps = 100000, 12, 30
        algo = algorithms.NSGA2(lb=torch.zeros(d, device=dev), ub=torch.ones(d, device=dev), n_objs=3, pop_size=pop)
        wf = workflows.StdWorkflow(algo, problems.numerical.DTLZ2(m=3))
        state, dt = timed(wf, wf.init(ex.PRNGKey(2).to(dev)), steps, 10)
        return {"config": f"NSGA-II pop={pop} DTLZ2 m=3 d={d}", "gens_per_sec": steps / dt, "ms_per_step": 1e3 * dt / steps,
                "evals_per_sec": pop * steps / dt, "steps": steps, "warmup": 10}

    def shade():
        pop, d, steps = 50000, 20, 200
        algo = algorithms.SHADE(lb=torch.full((d,), -100.0, device=dev), ub=torch.full((d,), 100.0, device=dev), pop_size=pop)
        # a generation is a dozen short launches: it is replayed from a HIP graph (same work; eager issue is host-bound);
        # "ring": the loop keeps no earlier state, so the replay hands out its own buffers instead of copies
        wf = workflows.StdWorkflow(algo, problems.numerical.CEC2022TestSuit.create(1), graph_step=True, graph_mode="ring")
        state, dt = timed(wf, wf.init(ex.PRNGKey(4).to(dev)), steps, 30)
        return {"config": f"SHADE pop={pop} CEC2022-F1 d={d} (one island), generation replayed from a HIP graph",
                "graph_replays": int(wf.graph_replays), "gens_per_sec": steps / dt,
                "ms_per_step": 1e3 * dt / steps, "evals_per_sec": pop * steps / dt, "steps": steps, "warmup": 30,
                "best": float(state.get_child_state("algorithm").fitness.min())}

    def openes():
        from evox_amd.models import MLP
        from evox_amd.problems.neuroevolution import SyntheticLocomotion, TensorEnvProblem

        obs_dim, act_dim, pop, horizon, steps = 244, 17, 4096, 50, 5
        net = MLP([obs_dim, 1024, 720, act_dim])
        prob = TensorEnvProblem(net.batched_apply, SyntheticLocomotion(obs_dim, act_dim, seed=4), max_episode_length=horizon,
                                num_episodes=1)
        algo = algorithms.OpenES(net.init(ex.PRNGKey(0)).to(dev), pop_size=pop, learning_rate=0.01, noise_stdev=0.05,
                                 optimizer="adam", mirrored_sampling=True)
        wf = workflows.StdWorkflow(algo, prob, opt_direction="max")
        state, dt = timed(wf, wf.init(ex.PRNGKey(3).to(dev)), steps, 2)
        return {"config": f"OpenES MLP {net.sizes} ({net.n_params} params) pop={pop}, synthetic locomotion obs={obs_dim}, "
                          f"episode_length={horizon}, fp32 parameters", "evals_per_sec": pop * steps / dt,
                "env_steps_per_sec": pop * horizon * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": 2}

    def openes_hopper():
        from evox_amd.models import MLP
        from evox_amd.problems.neuroevolution import Brax

        pop, horizon, steps = 4096, 100, 20
        net = MLP([11, 32, 32, 3])
        prob = Brax(net.batched_apply, "hopper", horizon, 1)          # in-tree planar physics, one lane per environment
        algo = algorithms.OpenES(net.init(ex.PRNGKey(0)).to(dev), pop_size=pop, learning_rate=0.03, noise_stdev=0.1,
                                 optimizer="adam", mirrored_sampling=True)
        wf = workflows.StdWorkflow(algo, prob, opt_direction="max")
        state, dt = timed(wf, wf.init(ex.PRNGKey(3).to(dev)), steps, 5)
        return {"config": f"OpenES MLP {net.sizes} pop={pop}, planar hopper (articulated-body physics, penalty contact), "
                          f"episode_length={horizon}", "evals_per_sec": pop * steps / dt,
                "env_steps_per_sec": pop * horizon * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": 5}

    out = {}
    for name, fn in (("nsga2_dtlz2_pop100k", nsga2), ("shade_cec2022_f1_pop50k", shade), ("openes_mlp_1m_pop4096", openes),
                     ("openes_hopper_pop4096", openes_hopper)):
        try:
            out[name] = fn()
        except Exception as e:               # the headline above stands on its own
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def main():
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:                 # the driver reads one JSON line whatever happens
        import traceback

        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "evaluations/sec, CMA-ES pop=10k dim=1k Ackley", "value": None,
                              "error": f"{type(e).__name__}: {e}"}), flush=True)
        raise SystemExit(1)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pop", type=int, default=10000, help="population per GPU (weak) or total (strong)")
    ap.add_argument("--dim", type=int, default=1000)
    ap.add_argument("--problem", default="ackley", choices=["ackley", "sphere", "rosenbrock", "ellipsoid"])
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"],
                    help="operands of the sampling and covariance GEMMs (accumulation, state and factorisation are fp32)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--eigh", default=None, help="eigensolver override: jacobi | solver")
    ap.add_argument("--decomp", default="newton_schulz", choices=["eigh", "newton_schulz"],
                    help="how C^(1/2), C^(-1/2) are obtained each generation: GEMM-only Newton-Schulz "
                         "iteration (default) or a full eigendecomposition (rocSOLVER)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: rehearsal of the multi-process path over gloo (not a benchmark)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a HIP graph")
    ap.add_argument("--prime", type=int, default=40,
                    help="untimed set-up generations on a throw-away state (GEMM plan timing, iteration control)")
    ap.add_argument("--resync-every", type=int, default=0,
                    help="multi-GPU: broadcast rank 0's replicated algorithm state every k generations (0 = never; the "
                         "replicas are computed from identical all-reduced statistics and differ at rounding level at most)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="single process: run the row-sharded workflow anyway (its cost without any communication)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="collective backend (default: nccl = RCCL on GPUs, gloo on CPU); gloo with --device cuda "
                         "rehearses the multi-rank path on a single GPU")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary configurations (NSGA-II 10^5, SHADE 50k, OpenES 4096, OpenES on the hopper) measured after the "
                         "headline on a single GPU")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start one fresh process per GPU (nothing in this process has touched
        # a GPU yet, and the children are started, never exec'ed into), relay rank 0's JSON line and return the
        # worst exit code
        raise SystemExit(_self_launch(args.gpus))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    on_gpu = args.device == "cuda"
    # counting devices does not initialise one: a job with more ranks than GPUs ends here, before any rank touches a device
    if on_gpu and (args.backend or "nccl") == "nccl" and world > 1 and world > torch.cuda.device_count():
        raise RuntimeError(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible: RCCL needs one GPU per "
                           "rank (use --backend gloo to rehearse several ranks on one GPU)")

    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows
    from evox_amd.core import distributed as dist

    if world > 1:
        dist.init_process_group(args.backend or ("nccl" if on_gpu else "gloo"))
    rank = dist.get_process_id()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if on_gpu:
        local = local % max(torch.cuda.device_count(), 1)
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")

    total_pop = args.pop * world if args.scaling == "weak" else args.pop
    key = ex.PRNGKey(20240611)
    center = ex.random.uniform(ex.random.fold_in(key, 1), (args.dim,), minval=-5.0, maxval=5.0).to(dev)
    algo = algorithms.CMAES(center_init=center, init_stdev=1.0, pop_size=total_pop, precision=args.precision,
                            eigh_method=args.eigh, decomposition=args.decomp)
    problem = {"ackley": problems.numerical.Ackley, "sphere": problems.numerical.Sphere,
               "rosenbrock": problems.numerical.Rosenbrock, "ellipsoid": problems.numerical.Ellipsoid}[args.problem]()
    if world > 1 or args.force_sharded:
        from evox_amd.workflows import ShardedWorkflow

        wf = ShardedWorkflow(algo, problem, resync_every=args.resync_every)
    else:
        wf = workflows.StdWorkflow(algo, problem, graph_step=args.graph, graph_mode="ring")
    state = wf.init(key)
    # One-time setup, the counterpart of a tracing compiler's compilation: the first generations time the candidate
    # GEMM algorithms of every shape (plans are cached afterwards) and let the factorisation's iteration control settle
    # from its conservative start.  Done on a throw-away state so that the measured run starts from the initial state
    # whatever --warmup is.
    scratch = state
    for _ in range(args.prime):
        scratch = wf.step(scratch)
    del scratch

    def sync():
        dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        state = wf.step(state)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = wf.step(state)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dt = float(dist.all_reduce(t, "max")[0])

    alg_state = state.get_child_state("algorithm")
    sigma = float(alg_state.sigma)
    best = float(problem.evaluate(None, alg_state.mean.reshape(1, -1).float())[0][0])
    ns = getattr(algo, "_ns", None)
    ns_report = None
    if ns is not None and "ns_iters" in alg_state:
        # which driver factorised, with how many iterations, and the condition bound it acted on
        ns_report = {"iterations": int(alg_state.ns_iters), "calls": int(alg_state.ns_calls),
                     "symmetric_bf16_calls": int(ns.symmetric_calls), "guard_fallbacks": int(ns.fallbacks),
                     "cond_estimate_below": float(alg_state.ns_res1[0]), "cond_bound_above": float(alg_state.ns_res1[1])}
    if rank == 0:
        ms = dt / args.steps * 1e3
        evals = total_pop * args.steps / dt
        out = {
            "metric": f"evaluations/sec, CMA-ES pop=10k dim=1k {args.problem.capitalize()}",
            "value": evals,
            "unit": "evaluations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "gens_per_sec": args.steps / dt,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": f"synthetic ({args.problem.capitalize()} objective, random initial mean; no data set)",
            "config": {"model": f"CMA-ES dim={args.dim} on {args.problem}", "global_batch": total_pop,
                       "pop_per_gpu": total_pop // world, "seq_len": args.dim,
                       "parallelism": f"dp{world} (row-sharded sampling/evaluation/covariance, replicated factorisation)",
                       "decomp_per_iter": algo.decomp_per_iter, "decomposition": args.decomp, "mu": algo.mu, "graph": bool(args.graph),
                       "setup_generations_before_warmup": args.prime,
                       "precision": ("bf16 operands in the sampling and covariance GEMMs and, as split hi+lo planes, in the "
                                     "Newton-Schulz factorisation products; fp32 accumulation, state and fitness"
                                     if args.precision == "bf16" else "fp32 throughout")},
            "sigma_end": sigma,
            "best_fitness_end": best,
            "factorisation": ns_report,
            "factorisation_residual": getattr(getattr(algo, "_ns", None), "last_residual", None),
        }
        if world == 1 and on_gpu and not args.no_secondary:
            del state, wf, algo
            torch.cuda.empty_cache()
            out["secondary"] = _secondary(dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as td

        td.destroy_process_group()


if __name__ == "__main__":
    main()
