"""Multi-process behaviour on CPU (gloo, 127.0.0.1): sharded / replicated / island workflows
give the single-process answer for world sizes 1, 2 and 4."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(case, world, tmp_path, extra_env=None):
    out = str(tmp_path / f"{case}_{world}.json")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), case, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def test_collectives(tmp_path):
    res = launch("collectives", 2, tmp_path)
    for r in res:
        assert r["gather"] == [0.0, 1.0, 2.0, 10.0, 11.0, 12.0]
        assert r["sum"] == [10.0, 12.0, 14.0] and r["max"] == [10.0, 11.0, 12.0]
        assert r["bcast"] == [10.0, 11.0, 12.0] and r["gather_axis1"] == [1, 6]
        assert r["begin_wait"] == r["inplace"] == r["sum"]


@pytest.mark.parametrize("case", ["cmaes", "openes", "pgpe", "ars", "des", "snes", "sepcmaes", "maes"])
def test_row_sharded_matches_single_process(case, tmp_path):
    base = launch(case, 1, tmp_path)[0]
    for world in (2, 4):
        res = launch(case, world, tmp_path)
        for r in res:
            assert r["row_sharded"] and r["world"] == world
            assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=2e-3, rtol=2e-3)
        assert all(r["vec"] == res[0]["vec"] for r in res), "replicas diverged"
        if case in ("cmaes", "sepcmaes"):
            assert res[0]["local_rows"] == 16 // world


@pytest.mark.parametrize("world", [2, 8])
def test_newton_schulz_replicas_stay_identical(world, tmp_path):
    """The factorisation's iteration control lives in the algorithm state and is a function of the run's history
    only: every rank takes the same decisions, so the replicated state stays bit-identical without any resync —
    also at the world size of a full node."""
    env = {"EVOX_TEST_DECOMP": "newton_schulz"}
    base = launch("cmaes", 1, tmp_path, env)[0]
    res = launch("cmaes", world, tmp_path, env)
    assert all(r["vec"] == res[0]["vec"] for r in res), "replicas diverged"
    assert res[0]["row_sharded"] and res[0]["local_rows"] == 16 // world
    assert torch.allclose(torch.tensor(res[0]["vec"]), torch.tensor(base["vec"]), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("decomp", ["eigh", "newton_schulz"])
def test_deterministic_row_sharded_cma_es_is_independent_of_the_world_size(decomp, tmp_path):
    """``CMAES(deterministic=True)``: fixed row blocks, their partial sums gathered and added in block order — mean and
    covariance after 30 generations are the same BITS on 1, 2, 4 and 8 processes."""
    env = {"EVOX_TEST_DECOMP": decomp}
    base = launch("cmaes_det", 1, tmp_path, env)[0]
    for world in (2, 4, 8):
        res = launch("cmaes_det", world, tmp_path, env)
        for r in res:
            assert r["row_sharded"] and r["world"] == world and r["local_rows"] == 16 // world
            assert r["vec"] == base["vec"] and r["cov"] == base["cov"]


def test_sharded_checkpoint_resume_and_reshard(tmp_path):
    """Two ranks write one checkpoint (replicated part once, row-local part per rank); resuming it reproduces the
    uninterrupted run bit for bit, and a four-rank job and a single process can pick the same checkpoint up."""
    env = {"EVOX_TEST_CKPT": str(tmp_path / "ckpt")}
    res = launch("sharded_ckpt", 2, tmp_path, env)
    assert all(r["same"] for r in res) and res[0]["rows"] == 8
    assert res[0]["files"] == ["manifest.json", "rank00000.safetensors", "rank00001.safetensors", "tensors.safetensors"]
    for world in (4, 1):
        again = launch("sharded_ckpt_load", world, tmp_path, env)
        assert again[0]["rows"] == 16 // world and again[0]["world"] == world
        assert torch.allclose(torch.tensor(again[0]["vec"]), torch.tensor(res[0]["vec"]), atol=2e-3, rtol=2e-3)


def test_resync_brings_a_drifted_replica_back(tmp_path):
    """``resync_every``: rank 1's covariance and step size are nudged at generation 12; after the broadcast at
    generation 15 the replicas are bit-identical again and stay so."""
    res = launch("cmaes_resync", 2, tmp_path)
    assert res[0]["row_sharded"]
    assert res[0]["vec"] == res[1]["vec"] and res[0]["cov"] == res[1]["cov"]


def test_row_sharded_pso_is_the_single_process_swarm_bit_for_bit(tmp_path):
    """Particles dealt out over the ranks: draws are addressed by (key, particle, variable) and every update is
    row-local, so the ranks' blocks put together ARE the single-process swarm — independent of the number of ranks, to
    the last bit."""
    base = launch("pso", 1, tmp_path)[0]
    for world in (2, 4, 8):
        res = launch("pso", world, tmp_path)
        assert all(r["row_sharded"] and r["world"] == world for r in res)
        rows = [row for r in sorted(res, key=lambda r: r["rank"]) for row in r["rows"]]
        assert len(res[0]["rows"]) == 16 // world
        assert rows == base["rows"]
        assert all(r["vec"] == base["vec"] for r in res)


def test_replicated_mode_for_algorithms_without_sharding(tmp_path):
    base = launch("pso_replicated", 1, tmp_path)[0]
    res = launch("pso_replicated", 2, tmp_path)
    assert not res[0]["row_sharded"]
    for r in res:
        assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=1e-5)


def test_enable_multi_devices(tmp_path):
    base = launch("std_multi_device", 1, tmp_path)[0]
    res = launch("std_multi_device", 2, tmp_path)
    assert res[0]["world"] == 2
    for r in res:
        assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=1e-5)


def test_clustered_algorithm_blocks_dealt_out_over_ranks(tmp_path):
    """Cross-process decomposition: 4 blocks on 2 ranks, two blocks per rank (as one batched CMA-ES instance), identical
    full candidates on both ranks, and the optimisation converges."""
    res = launch("clustered_dist", 2, tmp_path)
    assert all(r["members"] == 1 and r["blocks"] == 2 and r["cand_shape"] == [16, 20] for r in res)
    assert res[0]["cands"] == res[1]["cands"]
    assert all(r["best"] < 1.0 for r in res)


def test_island_model(tmp_path):
    res = launch("island", 2, tmp_path)
    assert res[0]["first"] != res[1]["first"], "islands must start from different populations"
    assert all(r["best"] < 1e-2 for r in res)


def test_process_distributed_workflow_cso():
    """The reference's Ray scenario (``tests/test_workflows.py:93-118``): CSO on Ackley over two
    workers, monitored from the driver."""
    import warnings

    import torch

    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows
    from evox_amd.monitors import StdSOMonitor

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monitor = StdSOMonitor()
    wf = workflows.RayDistributedWorkflow(
        algorithm=algorithms.CSO(lb=torch.full((2,), -32.0), ub=torch.full((2,), 32.0), pop_size=20),
        problem=problems.numerical.Ackley(), num_workers=2, monitors=[monitor], options={"num_cpus": 1})
    try:
        state = wf.init(ex.PRNGKey(42))
        for _ in range(100):
            state = wf.step(state)
        wf.flush(state)
        assert float(monitor.get_best_fitness()) < 1e-4
        assert len(monitor.get_history()) == 100
        cands, state = wf.sample(state)
        assert cands.shape[1] == 2
    finally:
        wf.close()


class _SplitSphere:
    """Sphere on the training split, shifted sphere on the validation split (the ``valid`` protocol of the
    reference's distributed workflow, ``workflows/distributed.py:145-153``)."""

    def __new__(cls):
        from evox_amd.core.problem import Problem
        from evox_amd.core.state import State

        class SplitSphere(Problem):
            def setup(self, key):
                return State(shift=torch.tensor(0.0))

            def valid(self, state, metric="loss"):
                return state.replace(shift=torch.tensor(1.0 if metric == "loss" else 2.0))

            def evaluate(self, state, X):
                return ((X - state.shift) ** 2).sum(dim=1), state

        return SplitSphere()


def test_process_distributed_workflow_metrics_valid_and_uneven_slices():
    """21 candidates over 2 workers (slices of 11 and 10), a driver-side monitor that receives the full candidate
    set, per-generation metrics, a global fitness transform, and ``valid`` on the validation split."""
    import evox_amd as ex
    from evox_amd import algorithms, workflows
    from evox_amd.monitors import EvalMonitor

    seen = {}

    class Probe(EvalMonitor):
        def post_eval(self, state, cands, tcands, fitness):
            seen["cands"], seen["fit"] = tuple(cands.shape), tuple(fitness.shape)
            seen["aligned"] = bool(torch.allclose((cands ** 2).sum(dim=1), fitness, atol=1e-4))
            super().post_eval(state, cands, tcands, fitness)

    mon = Probe()
    wf = workflows.RayDistributedWorkflow(
        algorithm=algorithms.PSO(lb=torch.full((3,), -5.0), ub=torch.full((3,), 5.0), pop_size=21),
        problem=_SplitSphere(), num_workers=2, monitors=[mon],
        metrics={"best": lambda f: f.min(), "mean": lambda f: f.mean()},
        global_fit_transform=[lambda f: f * 0.5], options={"num_cpus": 1})
    try:
        state = wf.init(ex.PRNGKey(7))
        for _ in range(12):
            state = wf.step(state, block=True)
        # hooks see the fitness before any transform, row-aligned with the full candidate set
        assert seen["cands"] == (21, 3) and seen["fit"] == (21,) and seen["aligned"]
        assert len(wf.metric_history["best"]) == 12 and float(wf.latest_metrics["best"]) <= float(wf.metric_history["best"][0])
        vfit, state = wf.valid(state, metric="loss")
        cands, state = wf.sample(state)
        assert vfit.shape == (21,)
        assert torch.allclose(vfit, ((cands - 1.0) ** 2).sum(dim=1), atol=1e-5)
    finally:
        wf.close()


def test_process_distributed_workflow_worker_failure_is_reported():
    """A worker that raises does not leave the driver blocked on the other workers' pipes."""
    import time

    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows

    wf = workflows.RayDistributedWorkflow(
        algorithm=algorithms.PSO(lb=torch.full((3,), -5.0), ub=torch.full((3,), 5.0), pop_size=20),
        problem=problems.numerical.Sphere(), num_workers=2, options={"num_cpus": 1})
    try:
        state = wf.init(ex.PRNGKey(7))
        state = wf.step(state, block=True)
        t0 = time.time()
        wf._send_all("fail", 1)
        with pytest.raises(RuntimeError, match="injected worker failure"):
            wf._recv_all()
        assert time.time() - t0 < 30
        assert wf._procs == []                      # everything it started is gone
    finally:
        wf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cmaes", "cmaes_wide", "openes", "pso"])
def test_row_sharded_on_gpu_matches_single_process(case, tmp_path):
    """The GPU code path of the row-sharded tell (gather kernel, Gram GEMM, fused path kernel): two
    and four ranks share the one GPU of the test box and talk over gloo."""
    env = {"EVOX_TEST_DEVICE": "cuda", "EVOX_TEST_DECOMP": "newton_schulz"}
    base = launch(case, 1, tmp_path, env)[0]
    for world in (2, 4):
        res = launch(case, world, tmp_path, env)
        for r in res:
            assert r["row_sharded"] and r["world"] == world
            assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=2e-3, rtol=2e-3)


@pytest.mark.gpu
def test_row_sharded_bf16_kernels_on_gpu_match_single_process(tmp_path):
    """The configuration of the multi-GPU benchmark: bf16 operands on the hand-written sampling / covariance kernels,
    Newton-Schulz factorisation, row-sharded over 2 and 4 ranks (sharing the one GPU, gloo).  bf16 rounding of the local
    covariance shares differs with the split, so the means agree to bf16-level tolerance, and the replicas exactly."""
    env = {"EVOX_TEST_DEVICE": "cuda", "EVOX_TEST_DECOMP": "newton_schulz", "EVOX_TEST_PRECISION": "bf16"}
    base = launch("cmaes_wide", 1, tmp_path, env)[0]
    for world in (2, 4):
        res = launch("cmaes_wide", world, tmp_path, env)
        for r in res:
            assert r["row_sharded"] and r["world"] == world
            assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=5e-2, rtol=5e-2)
        assert all(r["vec"] == res[0]["vec"] for r in res), "replicas diverged"


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_deterministic_row_sharded_cma_es_on_gpu(precision, tmp_path):
    """The deterministic mode on the GPU kernels (bf16: selection, gather, Gram product and the folded tail per fixed
    block; fp32: library products of one shape per block): 1, 2 and 4 ranks sharing the one GPU end with the same bits."""
    env = {"EVOX_TEST_DEVICE": "cuda", "EVOX_TEST_DECOMP": "newton_schulz", "EVOX_TEST_PRECISION": precision}
    base = launch("cmaes_det_wide", 1, tmp_path, env)[0]
    for world in (2, 4):
        res = launch("cmaes_det_wide", world, tmp_path, env)
        for r in res:
            assert r["row_sharded"] and r["world"] == world
            assert r["vec"] == base["vec"]


@pytest.mark.gpu
def test_row_sharded_cma_es_survives_a_fitness_of_ties(tmp_path):
    """All fitness values equal: the stable ranking hands every selected row to the lower rank, beyond the slots its
    selected rows are compacted into.  The update leaves the surplus out (finite state on both ranks) and warns once."""
    res = launch("cmaes_ties", 2, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    assert all(r["finite"] for r in res)
    assert res[0]["warned"] == 1 and res[1]["warned"] == 1


@pytest.mark.gpu
def test_resync_on_gpu_leaves_identical_replicas(tmp_path):
    """Two ranks on the one GPU (gloo, device tensors staged through the host): a nudged replica and whatever
    last-bit differences the ranks' library GEMMs produce are gone after the broadcast that ends generation 30."""
    res = launch("cmaes_resync", 2, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    assert res[0]["row_sharded"]
    assert res[0]["vec"] == res[1]["vec"] and res[0]["cov"] == res[1]["cov"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_one_shot_p2p_collectives(world, tmp_path):
    """all_gather / all_reduce through IPC-mapped receive buffers with flag hand-shakes, ranks sharing
    the one GPU of the test box (handles are exchanged over gloo)."""
    res = launch("p2p", world, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    assert all(r["ok"] for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_p2p_collectives_replay_from_a_graph(world, tmp_path):
    """The collectives' sequence number lives on the device, so their launches can be recorded: four replays per
    message size (one-shot and two-shot sums, all-gather), eager collectives and synchronisations in between, every
    replay checked."""
    res = launch("p2p_graph", world, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    assert all(r["ok"] for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_sharded_generation_replays_from_a_graph_with_p2p_collectives(world, tmp_path):
    """A whole row-sharded CMA-ES generation (sampling, evaluation, distributed ranking with its all-gather, the local
    sums, their all-reduce, the folded tail) recorded into a HIP graph whose collectives are the peer-to-peer kernels:
    2, 4 and 8 ranks sharing the one GPU replay it in step."""
    res = launch("sharded_p2p_graph", world, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    for r in res:
        # the swarm is bit for bit the eager one; CMA-ES polls its factorisation control under replay instead of adapting
        # it every generation, so its mean agrees to the accuracy of the factorisation
        assert r["pso"]["replays"] >= 10 and r["pso"]["graph"] == r["pso"]["eager"]
        assert r["cmaes"]["replays"] >= 2
        assert torch.allclose(torch.tensor(r["cmaes"]["graph"]), torch.tensor(r["cmaes"]["eager"]), atol=5e-2, rtol=5e-2)
    assert all(r["cmaes"]["graph"] == res[0]["cmaes"]["graph"] for r in res), "replicas diverged under replay"


@pytest.mark.gpu
def test_one_shot_p2p_wait_is_bounded(tmp_path):
    res = launch("p2p_timeout", 2, tmp_path, {"EVOX_TEST_DEVICE": "cuda"})
    assert res[0]["timed_out"] is True and res[0]["poisoned"] is True and res[0]["routed_raises"] is True
    assert res[1]["timed_out"] is False and res[1]["sum"] == [3.0] * 8


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cmaes", "openes"])
def test_row_sharded_over_rccl(case, tmp_path):
    """One rank per GPU over RCCL — the production path.  Needs at least two GPUs; the one-GPU test box skips it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {"EVOX_TEST_DEVICE": "cuda", "EVOX_TEST_DECOMP": "newton_schulz"}
    base = launch(case, 1, tmp_path, env)[0]
    res = launch(case, 2, tmp_path, dict(env, EVOX_TEST_BACKEND="nccl"))
    for r in res:
        assert r["row_sharded"] and r["world"] == 2
        assert torch.allclose(torch.tensor(r["vec"]), torch.tensor(base["vec"]), atol=2e-3, rtol=2e-3)
