"""Secondary benchmark configurations (the headline one is ``bench.py`` at the repository root).

    python benchmarks/run.py pso_quickstart            # PSO pop=100 dim=2 Ackley, CPU plumbing
    python benchmarks/run.py cmaes_sphere_bf16         # CMA-ES pop=10k dim=1k Sphere, bf16 sampling
    python benchmarks/run.py nsga2_dtlz2 --pop 100000  # NSGA-II on 3-objective DTLZ2
    python benchmarks/run.py openes_mlp                # OpenES, ~1M-parameter MLP policy, pop=4096
    python benchmarks/run.py openes_physics --task ant # OpenES, small MLP policies on an in-tree physics task
    python benchmarks/run.py island_shade              # island-model SHADE on CEC'22 F1, pop=50k per island

Multi-GPU configurations: ``python benchmarks/run.py <name> --gpus N`` starts one process per GPU itself
(``python -m torch.distributed.run --nproc-per-node N benchmarks/run.py <name> --gpus N`` works too).  Each run prints one JSON line.  Timed regions are bracketed by a
barrier and a device synchronisation; synthetic data and random initial weights throughout."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)


def _gpus_requested(argv) -> int:
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _gpus_requested(sys.argv[1:]) > 1:
    # one fresh process per GPU, started before this process imports torch (evox_amd/utils/launch.py by path)
    import importlib.util

    _spec = importlib.util.spec_from_file_location("_evox_launch", os.path.join(_ROOT, "evox_amd", "utils", "launch.py"))
    _mod = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_mod)
    raise SystemExit(_mod.launch_ranks(__file__, sys.argv[1:], _gpus_requested(sys.argv[1:])))

import torch                                                    # noqa: E402

import evox_amd as ex                                           # noqa: E402
from evox_amd import algorithms, problems, workflows          # noqa: E402
from evox_amd.core import distributed as dist                 # noqa: E402


def _sync(device):
    dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _timed(wf, state, steps, warmup, device):
    for _ in range(warmup):
        state = wf.step(state)
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        state = wf.step(state)
    _sync(device)
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dt = float(dist.all_reduce(t, "max")[0])
    return state, dt


def _setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("nccl" if args.device == "cuda" and torch.cuda.is_available() else "gloo")
    if args.device == "cuda" and torch.cuda.is_available():
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    return device, world, dist.get_process_id()


def _emit(rank, **kw):
    if rank == 0:
        print(json.dumps(kw), flush=True)


def pso_quickstart(args):
    device = torch.device("cpu")
    algo = algorithms.PSO(lb=torch.full((2,), -32.0), ub=torch.full((2,), 32.0), pop_size=100)
    wf = workflows.StdWorkflow(algo, problems.numerical.Ackley())
    state = wf.init(ex.PRNGKey(42))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    best = float(state.get_child_state("algorithm").global_best_fitness) if hasattr(state.get_child_state("algorithm"), "global_best_fitness") else None
    _emit(0, config="PSO pop=100 dim=2 Ackley (CPU)", gens_per_sec=args.steps / dt, ms_per_step=1e3 * dt / args.steps, best=best)


def cmaes_sphere_bf16(args):
    device, world, rank = _setup(args)
    n, lam = args.dim or 1000, args.pop or 10000
    algo = algorithms.CMAES(center_init=torch.full((n,), 3.0, device=device), init_stdev=1.0, pop_size=lam,
                            decomposition=args.decomp, precision="bf16")
    wf = workflows.StdWorkflow(algo, problems.numerical.Sphere())
    state = wf.init(ex.PRNGKey(1).to(device))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    a = state.get_child_state("algorithm")
    _emit(rank, config=f"CMA-ES pop={lam} dim={n} Sphere bf16 sampling, decomposition={args.decomp}", evals_per_sec=lam * args.steps / dt,
          gens_per_sec=args.steps / dt, ms_per_step=1e3 * dt / args.steps, sigma=float(a.sigma), dtype="bf16 operands / fp32 accumulate")


def nsga2_dtlz2(args):
    device, world, rank = _setup(args)
    pop, d = args.pop or 100000, args.dim or 12
    algo = algorithms.NSGA2(lb=torch.zeros(d, device=device), ub=torch.ones(d, device=device), n_objs=3, pop_size=pop)
    prob = problems.numerical.DTLZ2(m=3)
    wf = workflows.StdWorkflow(algo, prob)
    state = wf.init(ex.PRNGKey(2).to(device))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    from evox_amd.operators import non_dominated_sort
    fit = state.get_child_state("algorithm").fitness
    merged = torch.cat([fit, fit + 0.01 * torch.rand_like(fit)])
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(3):
        non_dominated_sort(merged)
    _sync(device)
    nd_ms = (time.perf_counter() - t0) / 3 * 1e3
    _emit(rank, config=f"NSGA-II pop={pop} DTLZ2 m=3 d={d}", gens_per_sec=args.steps / dt, ms_per_step=1e3 * dt / args.steps,
          evals_per_sec=pop * args.steps / dt, nd_sort_ms_2N=nd_ms)


def openes_mlp(args):
    device, world, rank = _setup(args)
    from evox_amd.models import MLP
    from evox_amd.problems.neuroevolution import SyntheticLocomotion, TensorEnvProblem
    obs_dim, act_dim = 244, 17                                   # humanoid-sized observation / action
    net = MLP([obs_dim, 1024, 720, act_dim])                     # ≈1.0 M parameters
    pop = args.pop or 4096
    dtype = torch.bfloat16 if args.param_dtype == "bf16" else torch.float32
    policy = net.batched_apply
    from evox_amd.ops.es import cast_population
    to_param_dtype = [] if dtype == torch.float32 else [lambda w: cast_population(w, dtype)]     # once per generation
    prob = TensorEnvProblem(policy, SyntheticLocomotion(obs_dim, act_dim, seed=4), max_episode_length=args.episode_length,
                            num_episodes=1)
    center = net.init(ex.PRNGKey(0)).to(device)
    cls = algorithms.PGPE if args.algo == "pgpe" else algorithms.OpenES
    if args.algo == "pgpe":
        algo = cls(pop_size=pop, center_init=center, optimizer="adam", stdev_init=0.05)
    else:
        algo = cls(center, pop_size=pop, learning_rate=0.01, noise_stdev=0.05, optimizer="adam", mirrored_sampling=True)
    if world > 1:
        wf = workflows.ShardedWorkflow(algo, prob, opt_direction="max", candidate_transforms=to_param_dtype)
    else:
        wf = workflows.StdWorkflow(algo, prob, opt_direction="max", candidate_transforms=to_param_dtype)
    state = wf.init(ex.PRNGKey(3).to(device))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    _emit(rank, config=f"{cls.__name__} MLP {net.sizes} ({net.n_params} params) pop={pop} x{world} ranks, synthetic locomotion "
          f"obs={obs_dim}, episode_length={args.episode_length}, params {args.param_dtype}",
          evals_per_sec=pop * args.steps / dt, env_steps_per_sec=pop * args.episode_length * args.steps / dt,
          gens_per_sec=args.steps / dt, ms_per_step=1e3 * dt / args.steps, n_gpus=world)


def openes_physics(args):
    """OpenES over small MLP policies on an in-tree physics task (``--task hopper|walker2d|halfcheetah|swimmer|ant|…``):
    the environment step, its observation, reward, termination and the return accumulation are one launch per step."""
    device, world, rank = _setup(args)
    from evox_amd.models import MLP
    from evox_amd.problems.neuroevolution import Brax, make_tensor_env
    env = make_tensor_env(args.task)
    net = MLP([env.obs_dim, 32, 32, env.act_dim])
    pop = args.pop or 4096
    prob = Brax(net.batched_apply, args.task, args.episode_length, 1)
    algo = algorithms.OpenES(net.init(ex.PRNGKey(0)).to(device), pop_size=pop, learning_rate=0.03, noise_stdev=0.1,
                             optimizer="adam", mirrored_sampling=True)
    cls = workflows.ShardedWorkflow if world > 1 else workflows.StdWorkflow
    kw = {} if world > 1 else {"graph_step": args.graph}
    wf = cls(algo, prob, opt_direction="max", **kw)
    state = wf.init(ex.PRNGKey(3).to(device))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    _emit(rank, config=f"OpenES MLP {net.sizes} pop={pop} x{world} ranks, {args.task} (in-tree physics), "
          f"episode_length={args.episode_length}", evals_per_sec=pop * args.steps / dt,
          env_steps_per_sec=pop * args.episode_length * args.steps / dt, gens_per_sec=args.steps / dt,
          ms_per_step=1e3 * dt / args.steps, n_gpus=world)


def island_shade(args):
    device, world, rank = _setup(args)
    pop, d = args.pop or 50000, args.dim or 20
    algo = algorithms.SHADE(lb=torch.full((d,), -100.0, device=device), ub=torch.full((d,), 100.0, device=device), pop_size=pop)
    prob = problems.numerical.CEC2022TestSuit.create(1)
    wf = workflows.IslandWorkflow(algo, prob, migrate_every=args.migrate_every, num_migrants=args.migrants,
                                  graph_step=args.graph, graph_mode="ring")
    state = wf.init(ex.PRNGKey(4).to(device))
    state, dt = _timed(wf, state, args.steps, args.warmup, device)
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(5):
        state = wf._migrate(state)
    _sync(device)
    mig_us = (time.perf_counter() - t0) / 5 * 1e6
    best = float(state.get_child_state("algorithm").fitness.min())
    _emit(rank, config=f"island SHADE CEC2022-F1 d={d}, {world} islands x pop={pop}, graph={args.graph}", evals_per_sec=world * pop * args.steps / dt,
          gens_per_sec=args.steps / dt, ms_per_step=1e3 * dt / args.steps, migration_us=mig_us, best=best, n_gpus=world)


CONFIGS = {f.__name__: f for f in (pso_quickstart, cmaes_sphere_bf16, nsga2_dtlz2, openes_mlp, openes_physics, island_shade)}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pop", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--gpus", type=int, default=1, help="ranks on this node (one per GPU); >1 self-launches")
    ap.add_argument("--decomp", default="newton_schulz")
    ap.add_argument("--algo", default="openes")
    ap.add_argument("--param-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--episode-length", type=int, default=50)
    ap.add_argument("--task", default="hopper", help="openes_physics: the physics task")
    ap.add_argument("--migrate-every", type=int, default=10)
    ap.add_argument("--migrants", type=int, default=64)
    ap.add_argument("--graph", action="store_true", help="replay steady-state generations from a HIP graph")
    args = ap.parse_args()
    CONFIGS[args.config](args)
    if dist.is_dist_initialized():
        import torch.distributed as td
        td.destroy_process_group()
