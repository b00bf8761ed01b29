"""Worker of the multi-process tests: run by tests/test_distributed.py once per rank with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (gloo, CPU)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import evox_amd as ex
from evox_amd import algorithms as A
from evox_amd import monitors, problems, workflows
from evox_amd.core import distributed as dist
from evox_amd.parallel import IslandWorkflow, ShardedWorkflow


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    backend = os.environ.get("EVOX_TEST_BACKEND", "gloo")
    if backend == "nccl":
        # RCCL: one GPU per rank
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dist.init_process_group(backend)
    rank, world = dist.get_process_id(), dist.get_world_size()
    dim = 8
    # EVOX_TEST_DEVICE=cuda: every rank computes on the one visible GPU while the collectives stay
    # on gloo (device data is staged through host memory by evox_amd.core.distributed)
    dev = torch.device(os.environ.get("EVOX_TEST_DEVICE", "cpu"))
    if backend == "nccl":
        dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    c = torch.full((dim,), 3.0, device=dev)
    lb, ub = torch.full((dim,), -10.0, device=dev), torch.full((dim,), 10.0, device=dev)
    result = {}
    if case in ("cmaes", "cmaes_det", "cmaes_det_wide", "cmaes_resync", "cmaes_wide", "openes", "pgpe", "pso_replicated", "pso", "ars", "des", "snes", "sepcmaes", "maes"):
        algo = {"cmaes_resync": lambda: A.CMAES(c, 1.0, pop_size=16),
                "cmaes": lambda: A.CMAES(c, 1.0, pop_size=16, decomposition=os.environ.get("EVOX_TEST_DECOMP", "eigh")),
                "cmaes_det": lambda: A.CMAES(c, 1.0, pop_size=16, decomposition=os.environ.get("EVOX_TEST_DECOMP", "eigh"),
                                             deterministic=True),
                "cmaes_det_wide": lambda: A.CMAES(c, 1.0, pop_size=2048, decomposition=os.environ.get("EVOX_TEST_DECOMP", "eigh"),
                                                  precision=os.environ.get("EVOX_TEST_PRECISION", "fp32"), deterministic=True),
                # local shares wide enough that only the rank-ordered head of the local rows is processed
                "cmaes_wide": lambda: A.CMAES(c, 1.0, pop_size=2048, decomposition=os.environ.get("EVOX_TEST_DECOMP", "eigh"),
                                              precision=os.environ.get("EVOX_TEST_PRECISION", "fp32")),
                "openes": lambda: A.OpenES(c, 16, 0.05, 0.1),
                "ars": lambda: A.ARS(16, c, elite_ratio=0.5, lr=0.1, sigma=0.1), "des": lambda: A.DES(16, c, sigma_init=0.5),
                "snes": lambda: A.SNES(16, c, sigma=0.5),
                "sepcmaes": lambda: A.SepCMAES(c, 1.0, pop_size=16), "maes": lambda: A.MAES(c, 1.0, pop_size=16),
                "pgpe": lambda: A.PGPE(16, c, "adam"), "pso": lambda: A.PSO(lb, ub, 16),
                # an algorithm without the row-sharded protocol: replicated state, sharded evaluation
                "pso_replicated": lambda: A.CSO(lb, ub, 16)}[case]()
        mon = monitors.EvalMonitor()
        wf = ShardedWorkflow(algo, problems.numerical.Ackley(), monitors=[mon],
                             resync_every=5 if case == "cmaes_resync" else 0)
        state = wf.init(ex.PRNGKey(3))
        for g in range(30):
            state = wf.step(state)
            if case == "cmaes_resync" and g == 11 and rank == 1:
                # a replica that has drifted: the next broadcast (generation 15) must bring it back
                a = state.get_child_state("algorithm")
                state = state.replace_child("algorithm", a.replace(C=a.C * 1.001, sigma=a.sigma * 1.001))
        alg = state.get_child_state("algorithm")
        vec = alg.mean if case in ("cmaes", "cmaes_det", "cmaes_det_wide", "cmaes_resync", "cmaes_wide", "sepcmaes", "maes") else (alg.global_best_location if case == "pso" else (alg.population[0] if case == "pso_replicated" else alg.center))
        result = {"vec": vec.tolist(), "world": state.world_size, "rank": state.rank,
                  "row_sharded": wf.row_sharded, "best": float(mon.get_best_fitness()),
                  "local_rows": int(alg.population.shape[0]) if case in ("cmaes", "cmaes_det", "cmaes_det_wide", "cmaes_resync", "cmaes_wide", "sepcmaes") else None,
                  "cov": alg.C.reshape(-1).tolist() if case in ("cmaes_resync", "cmaes_det") else None,
                  "rows": alg.population.tolist() if case == "pso" else None}
    elif case == "sharded_ckpt":
        path = os.environ["EVOX_TEST_CKPT"]

        def make():
            return ShardedWorkflow(A.CMAES(c, 1.0, pop_size=16, decomposition="newton_schulz"), problems.numerical.Rosenbrock())

        wf = make()
        state = wf.init(ex.PRNGKey(3))
        for _ in range(6):
            state = wf.step(state)
        wf.save_checkpoint(state, path)
        for _ in range(6):
            state = wf.step(state)
        wf2 = make()
        resumed = wf2.load_checkpoint(path, template=wf2.init(ex.PRNGKey(77)))
        rows = int(resumed.get_child_state("algorithm").population.shape[0])
        for _ in range(6):
            resumed = wf2.step(resumed)
        a, b = state.get_child_state("algorithm"), resumed.get_child_state("algorithm")
        result = {"same": bool(torch.equal(a.mean, b.mean) and torch.equal(a.C, b.C) and torch.equal(a.sigma, b.sigma)),
                  "rows": rows, "files": sorted(os.listdir(path)), "vec": a.mean.tolist()}
    elif case == "sharded_ckpt_load":
        # a job of another size picks the checkpoint up
        path = os.environ["EVOX_TEST_CKPT"]
        wf = ShardedWorkflow(A.CMAES(c, 1.0, pop_size=16, decomposition="newton_schulz"), problems.numerical.Rosenbrock())
        state = wf.load_checkpoint(path, template=wf.init(ex.PRNGKey(5)))
        rows = int(state.get_child_state("algorithm").population.shape[0])
        for _ in range(6):
            state = wf.step(state)
        result = {"rows": rows, "vec": state.get_child_state("algorithm").mean.tolist(), "world": state.world_size}
    elif case == "clustered_dist":
        # blocks dealt out over the ranks: each rank evolves its share, candidates are gathered along the variable axis
        base = A.CMAES(torch.full((5,), 3.0, device=dev), init_stdev=2.0, pop_size=16)
        algo = A.ClusterdAlgorithm(base, dim=20, num_cluster=4, distribute=True)
        mon = monitors.EvalMonitor()
        wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), monitors=[mon])
        state = wf.enable_multi_devices(wf.init(ex.PRNGKey(5)))
        cands, _ = wf.algorithm.init_ask(state.get_child_state("algorithm"))
        for _ in range(60):
            state = wf.step(state)
        result = {"members": len(algo.members), "blocks": algo.members[0].num_blocks, "cand_shape": list(cands.shape),
                  "best": float(mon.get_best_fitness()),
                  "cands": cands.reshape(-1)[:40].tolist()}
    elif case == "std_multi_device":
        wf = workflows.StdWorkflow(A.PSO(lb, ub, 16), problems.numerical.Sphere())
        state = wf.enable_multi_devices(wf.init(ex.PRNGKey(5)))
        for _ in range(20):
            state = wf.step(state)
        alg = state.get_child_state("algorithm")
        result = {"vec": alg.global_best_location.tolist(), "world": state.world_size}
    elif case == "island":
        mon = monitors.EvalMonitor()
        wf = IslandWorkflow(A.SHADE(lb, ub, 32), problems.numerical.Sphere(), monitors=[mon], migrate_every=5,
                            num_migrants=4)
        state = wf.init(ex.PRNGKey(9))
        first = state.get_child_state("algorithm").population[0].tolist()
        for _ in range(40):
            state = wf.step(state)
        alg = state.get_child_state("algorithm")
        result = {"first": first, "best": float(alg.fitness.min())}
    elif case == "collectives":
        x = torch.arange(3, dtype=torch.float32) + 10 * rank
        g = dist.all_gather(x)
        s = dist.all_reduce(x)
        m = dist.all_reduce(x, "max")
        b = dist.broadcast(x, src=world - 1)
        g2 = dist.all_gather(x.reshape(1, 3), axis=1)
        begun = dist.all_reduce_begin(x.clone()).wait()            # the split form gives what the in-place one gives
        inplace = dist.all_reduce_(x.clone())
        result = {"gather": g.tolist(), "sum": s.tolist(), "max": m.tolist(), "bcast": b.tolist(),
                  "gather_axis1": list(g2.shape), "begin_wait": begun.tolist(), "inplace": inplace.tolist()}
    elif case == "p2p":
        # one-shot collectives over IPC-mapped device memory: the ranks share the one visible GPU
        from evox_amd.parallel.p2p import P2PGroup

        torch.cuda.set_device(0)
        g = P2PGroup(max_bytes=1 << 20, timeout_s=10.0)
        ok = True
        gen = torch.Generator().manual_seed(1234)
        g.two_shot_min_bytes = 4096 * 4
        for it, n in enumerate([3, 1000, 4096, 65536, 7, 262144, 12, 1001, 4100, 65537]):
            full = torch.randn(world, n, generator=gen)              # same on every rank
            mine = full[rank].to(dev)
            got = g.all_gather(mine)
            ok &= bool(torch.equal(got.cpu(), full.reshape(-1)))
            red = g.all_reduce(mine)
            ok &= bool(torch.allclose(red.cpu(), full.sum(0), rtol=1e-5, atol=1e-5))
            ints = g.all_gather(torch.full((5,), rank + 10 * it, dtype=torch.int32, device=dev))
            ok &= ints.cpu().tolist() == [r + 10 * it for r in range(world) for _ in range(5)]
        g.check()
        # routed through the framework's collectives
        os.environ["EVOX_AMD_P2P"] = "1"
        x = torch.arange(6, dtype=torch.float32, device=dev) + 100 * rank
        ok &= dist.all_gather(x).cpu().tolist() == [float(v + 100 * r) for r in range(world) for v in range(6)]
        ok &= dist.all_reduce(x).cpu().tolist() == [float(world * v + 100 * sum(range(world))) for v in range(6)]
        os.environ["EVOX_AMD_P2P"] = "0"
        g.close()
        result = {"ok": ok}
    elif case == "p2p_graph":
        # the collectives recorded into a graph: the sequence number lives on the device, so every replay is a new
        # collective; replays are interleaved with eager collectives and synchronisations and each one is checked
        from evox_amd.parallel.p2p import P2PGroup

        torch.cuda.set_device(0)
        g = P2PGroup(max_bytes=1 << 20, timeout_s=10.0)
        g.two_shot_min_bytes = 4096 * 4
        ok = True
        gen = torch.Generator().manual_seed(99)
        for n in (37, 8192):                                   # one-shot and two-shot sums
            x = torch.zeros(n, device=dev)
            x.copy_(torch.randn(world, n, generator=gen)[rank])
            g.all_reduce(x)                                    # an eager collective before the recording
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                red = g.all_reduce(x)
                gat = g.all_gather(x)
            for it in range(4):
                full = torch.randn(world, n, generator=gen)
                x.copy_(full[rank])
                graph.replay()
                torch.cuda.synchronize()
                ok &= bool(torch.allclose(red.cpu(), full.sum(0), rtol=1e-5, atol=1e-5))
                ok &= bool(torch.equal(gat.cpu(), full.reshape(-1)))
                eager = g.all_reduce(x)                        # eager work between the replays
                ok &= bool(torch.allclose(eager.cpu(), full.sum(0), rtol=1e-5, atol=1e-5))
        g.check()
        dist.barrier()
        g.close()
        result = {"ok": ok}
    elif case == "cmaes_ties":
        # every individual gets the same fitness: ties go to the lower row index, so the lower ranks hold ALL selected
        # rows — more than their compaction slots.  The surplus is left out, the run goes on, and a warning says so.
        import warnings

        from evox_amd.core.problem import Problem

        class Flat(Problem):
            def evaluate(self, state, x):
                return torch.zeros(x.shape[0], device=x.device), state

        algo = A.CMAES(c, 1.0, pop_size=2048, decomposition="newton_schulz", precision="bf16")
        wf = ShardedWorkflow(algo, Flat())
        state = wf.init(ex.PRNGKey(3))
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(12):
                state = wf.step(state)
            torch.cuda.synchronize() if dev.type == "cuda" else None
        alg = state.get_child_state("algorithm")
        result = {"finite": bool(torch.isfinite(alg.mean).all() and torch.isfinite(alg.C).all()),
                  "warned": sum("left out" in str(w.message) for w in caught)}
    elif case == "sharded_p2p_graph":
        # a row-sharded generation replayed from a HIP graph, its collectives being the peer-to-peer kernels
        torch.cuda.set_device(0)
        os.environ["EVOX_AMD_P2P"] = "1"
        result = {}
        for name in ("pso", "cmaes"):
            vecs = {}
            for mode in (False, True):
                algo = A.PSO(lb, ub, 2048) if name == "pso" else A.CMAES(c, 1.0, pop_size=2048, decomposition="newton_schulz",
                                                                         precision="bf16")
                wf = ShardedWorkflow(algo, problems.numerical.Ackley(), graph_step=mode)
                state = wf.init(ex.PRNGKey(3))
                for _ in range(30):
                    state = wf.step(state)
                torch.cuda.synchronize()
                dist.check_p2p()
                alg = state.get_child_state("algorithm")
                vec = alg.global_best_location if name == "pso" else alg.mean
                vecs[mode] = (vec.tolist(), wf.graph_replays if mode else 0)
                dist.barrier()
            result[name] = {"eager": vecs[False][0], "graph": vecs[True][0], "replays": vecs[True][1]}
        os.environ["EVOX_AMD_P2P"] = "0"
    elif case == "p2p_timeout":
        # rank 1 joins the collective late; rank 0 polls a few times only, so its wait kernel gives up,
        # finishes, and the host reports the time-out (no wave is left spinning)
        import time

        from evox_amd.parallel.p2p import P2PGroup, P2PTimeout

        torch.cuda.set_device(0)
        g = P2PGroup(max_bytes=4096, timeout_s=10.0)
        if rank == 0:
            g._max_polls = 50
        else:
            time.sleep(2.0)
        x = torch.ones(8, device=dev) * (rank + 1)
        out = g.all_reduce(x)
        timed_out = False
        try:
            g.check()
        except P2PTimeout:
            timed_out = True
        poisoned = bool(torch.isnan(out).all()) if timed_out else False
        # the framework's own collectives on the same situation: the next routed call raises
        from evox_amd.parallel import p2p as p2p_mod

        p2p_mod._default = g
        os.environ["EVOX_AMD_P2P"] = "1"
        routed_raises = False
        if rank == 0:                      # rank 1 saw no time-out; a further collective would wait for rank 0 in vain
            try:
                dist.all_reduce(x)
            except P2PTimeout:
                routed_raises = True
        os.environ["EVOX_AMD_P2P"] = "0"
        p2p_mod._default = None
        result = {"timed_out": timed_out, "sum": out.cpu().tolist() if not timed_out else None, "poisoned": poisoned,
                  "routed_raises": routed_raises}
        dist.barrier()
        g.close()
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(result, f)
    dist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
