"""Container algorithms; scenarios of the reference's ``tests/test_containers.py:8-150``
(its CMA-ES and random-mask cases are skipped upstream as unstable; they run here)."""
import pytest
import torch

import evox_amd as ex
from evox_amd import Stateful, algorithms, problems, workflows
from evox_amd.monitors import EvalMonitor


def _run(algorithm, steps, problem=None):
    monitor = EvalMonitor()
    wf = workflows.StdWorkflow(algorithm, problems.numerical.Ackley() if problem is None else problem, monitors=[monitor])
    state = wf.init(ex.PRNGKey(42))
    for _ in range(steps):
        state = wf.step(state)
    return float(monitor.get_best_fitness()), state


def test_clustered_cma_es():
    algo = algorithms.ClusterdAlgorithm(
        base_algorithm=algorithms.CMAES(torch.full((10,), -10.0), init_stdev=6, pop_size=10), dim=40, num_cluster=4)
    best, _ = _run(algo, 300)
    assert best < 3


def test_clustered_separable_base_runs_as_one_fused_instance():
    """PSO / CSO treat every decision variable on its own, so all blocks together are one wider instance: no member loop,
    and the same optimisation behaviour as the loop."""
    results = {}
    for fuse in (False, "auto"):
        base = algorithms.CSO(lb=torch.full((10,), -32.0), ub=torch.full((10,), 32.0), pop_size=100)
        algo = algorithms.ClusterdAlgorithm(base_algorithm=base, dim=80, num_cluster=8, fuse=fuse)
        assert algo.fused == (fuse == "auto") and len(algo.members) == (1 if algo.fused else 8)
        results[fuse], state = _run(algo, 250)
        pop = state.get_child_state("algorithm").get_child_state("members").get_child_state("m0000").population
        assert pop.shape == (100, 80 if algo.fused else 10)
    assert results["auto"] < 6 and results[False] < 6
    with pytest.raises(TypeError):
        algorithms.ClusterdAlgorithm(algorithms.MAES(torch.zeros(10), init_stdev=1.0, pop_size=10), dim=40,
                                     num_cluster=4, fuse=True)             # neither separable nor available batched


@pytest.mark.parametrize("name", ["OpenES", "PGPE", "SNES", "DES", "ARS", "ESMC", "PersistentES", "Noise_reuse_es"])
def test_clustered_separable_es_strategies_fuse(name):
    """Isotropic / diagonal strategies with a per-variable optimizer are separable over blocks told the same fitness: the
    container runs ONE wide instance (hyper-parameters derived from the block dimension keep the member's values), and
    that instance optimises like the member loop does."""
    make = {"OpenES": lambda c: algorithms.OpenES(c, 64, 0.05, 0.1),
            "PGPE": lambda c: algorithms.PGPE(64, c, "adam", stdev_init=0.1),
            "SNES": lambda c: algorithms.SNES(64, c), "DES": lambda c: algorithms.DES(64, c),
            "ARS": lambda c: algorithms.ARS(64, c), "ESMC": lambda c: algorithms.ESMC(65, c, lr=0.1, sigma=0.1),
            "PersistentES": lambda c: algorithms.PersistentES(64, c, lr=0.1, sigma=0.1, T=10, K=10),
            "Noise_reuse_es": lambda c: algorithms.Noise_reuse_es(64, c, lr=0.1, sigma=0.1)}[name]
    best = {}
    for fuse in (False, "auto"):
        base = make(torch.full((5,), 2.0))
        algo = algorithms.ClusterdAlgorithm(base, dim=40, num_cluster=8, fuse=fuse)
        assert algo.fused == (fuse == "auto") and len(algo.members) == (1 if algo.fused else 8)
        assert base.dim == 5                                   # the caller's instance is left as it was
        best[fuse], state = _run(algo, 80, problems.numerical.Sphere())
        center = state.get_child_state("algorithm").get_child_state("members").get_child_state("m0000").center
        assert center.shape == (40 if algo.fused else 5,)
    start = 40 * 4.0
    assert best["auto"] < 0.5 * start and best[False] < 0.5 * start
    assert best["auto"] < 4 * best[False] + 1.0 and best[False] < 4 * best["auto"] + 1.0


def test_member_copies_own_their_sub_modules():
    """A strategy with a stateful optimizer inside (PGPE with ClipUp couples the variables, so it stays a member loop) is
    copied together with that optimizer: every member finds its own optimizer state."""
    base = algorithms.PGPE(32, torch.full((5,), 2.0), "clipup", stdev_init=0.1)
    algo = algorithms.ClusterdAlgorithm(base, dim=20, num_cluster=4)
    assert not algo.fused and len({id(m.optimizer) for m in algo.members}) == 4
    best, _ = _run(algo, 40, problems.numerical.Sphere())
    assert best < 80.0


@pytest.mark.parametrize("per_block_fitness", [False, True])
def test_batched_cma_es_follows_the_member_update(per_block_fitness):
    """One batched instance = S members: fed the same candidates and fitness, every block's mean, step size, paths and
    covariance follow the member's update, generation after generation (decomposition every second generation)."""
    from evox_amd.algorithms.so.es_variants.batched_cma import BatchedCMAES

    S, d, P = 3, 6, 12
    member = algorithms.CMAES(torch.linspace(-1, 1, d), init_stdev=0.7, pop_size=P, decomp_per_iter=2)
    batched = BatchedCMAES.of(member, S, concat=False)
    assert (batched.mu, batched.decomp_per_iter) == (member.mu, member.decomp_per_iter)
    for name in ("cc", "cs", "c1", "cmu", "damps", "chiN", "mueff"):
        assert getattr(batched, name) == pytest.approx(getattr(member, name), rel=1e-12)
    bstate = batched.init(ex.PRNGKey(3))
    mstates = [member.init(ex.PRNGKey(100 + s)) for s in range(S)]
    target = torch.randn(S, d, generator=torch.Generator().manual_seed(0))
    for gen in range(7):
        blocks, bstate = batched.ask(bstate)
        assert blocks.shape == (S, P, d)
        per_block = ((blocks - target[:, None, :]) ** 2).sum(-1) + 0.1 * torch.sin(5 * blocks).sum(-1)     # (S, P)
        fitness = per_block if per_block_fitness else per_block.sum(0)
        bstate = batched.tell(bstate, fitness)
        for s in range(S):
            _, ms = member.ask(mstates[s])
            ms = member.tell(ms.replace(population=blocks[s].clone()), fitness[s] if per_block_fitness else fitness)
            mstates[s] = ms
            for name in ("mean", "sigma", "ps", "pc", "C", "invsqrtC"):
                torch.testing.assert_close(getattr(bstate, name)[s], getattr(ms, name), rtol=2e-4, atol=2e-5, msg=f"{name} gen {gen}")
            recon = (ms.B * ms.D ** 2) @ ms.B.T
            brecon = (bstate.B[s] * bstate.D[s] ** 2) @ bstate.B[s].T
            torch.testing.assert_close(brecon, recon, rtol=2e-4, atol=2e-5)
    assert bstate.count_eigen == mstates[0].count_eigen == 3


@pytest.mark.parametrize("name", ["DE", "ODE", "JaDE", "SHADE", "CoDE"])
def test_blocked_de_family_follows_the_member_update(name):
    """One blocked instance = B members told the same fitness: given the instance's trial vectors and control
    parameters block by block, every member's own tell() reproduces the block's population and adaptation memory,
    generation after generation."""
    B, d, P = 3, 4, 24
    lb, ub = torch.linspace(-3, -1, d), torch.linspace(1, 3, d)
    member = {"DE": lambda: algorithms.DE(lb, ub, P, batch_size=8), "JaDE": lambda: algorithms.JaDE(lb, ub, P),
              "ODE": lambda: algorithms.ODE(lb, ub, P, batch_size=8), "CoDE": lambda: algorithms.CoDE(lb, ub, P),
              "SHADE": lambda: algorithms.SHADE(lb, ub, P, memory_size=5)}[name]()
    blocked = member.blocked(B)
    assert (blocked.blocks, blocked.dim, member.blocks, member.dim) == (B, B * d, 1, d)
    assert torch.equal(blocked.lb, lb.repeat(B))
    bstate = blocked.init(ex.PRNGKey(5))
    mstates = [member.init(ex.PRNGKey(50 + s)).replace(population=bstate.population.reshape(P, B, d)[:, s].clone())
               for s in range(B)]
    target = torch.randn(B * d, generator=torch.Generator().manual_seed(1)) * 0.5
    for gen in range(12):
        trial, bstate = blocked.ask(bstate)
        rows = trial.shape[0]
        assert trial.shape == ({"DE": 8, "ODE": 8, "CoDE": 3 * P}.get(name, P), B * d)
        assert bool((trial >= blocked.lb).all() and (trial <= blocked.ub).all())
        if name == "ODE" and gen % 2 == 1:                # opposition generation: every block mirrors an individual of its own
            mirrored = (blocked.ub + blocked.lb - trial).reshape(rows, B, d)
            blocks_pop = bstate.population.reshape(P, B, d)
            for s in range(B):
                assert all(bool(torch.isclose(blocks_pop[:, s], mirrored[r, s], atol=1e-5).all(1).any()) for r in range(rows))
        fitness = ((trial - target) ** 2).sum(1)
        after = blocked.tell(bstate, fitness)
        for s in range(B):
            ms = mstates[s].replace(trial_vectors=trial.reshape(rows, B, d)[:, s].clone(), start_index=bstate.start_index,
                                    start_index_dev=bstate.start_index_dev.clone())
            if name == "ODE":
                ms = ms.replace(counter=bstate.counter)
            if name in ("JaDE", "SHADE"):
                ms = ms.replace(F_vect=bstate.F_vect.reshape(P, B)[:, s].clone(), CR_vect=bstate.CR_vect.reshape(P, B)[:, s].clone())
            ms = member.tell(ms, fitness)
            mstates[s] = ms
            assert torch.equal(after.population.reshape(P, B, d)[:, s], ms.population), (gen, s)
            assert torch.equal(after.fitness, ms.fitness) and int(after.best_index) == int(ms.best_index)
            if name == "JaDE":
                torch.testing.assert_close(after.F_u[s], ms.F_u, rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(after.CR_u[s], ms.CR_u, rtol=1e-5, atol=1e-6)
            if name == "SHADE":
                torch.testing.assert_close(after.Memory_FCR[:, s], ms.Memory_FCR, rtol=1e-5, atol=1e-6)
        bstate = after
    if name == "SHADE":
        mem = bstate.Memory_FCR
        assert mem.shape == (2, B, 5) and not torch.equal(mem[:, 0], mem[:, 1])      # the blocks adapt on their own


def test_blocked_sep_cma_es_follows_the_member_update():
    """One blocked SepCMAES = B members: fed the instance's candidates block by block and the shared fitness, every
    member's mean, step size, paths and variances follow the block's."""
    B, d, P = 3, 5, 14
    member = algorithms.SepCMAES(torch.linspace(-1, 1, d), 0.8, pop_size=P)
    blocked = member.blocked(B)
    assert (blocked.blocks, blocked.dim, blocked.block_dim, member.blocks) == (B, B * d, d, 1)
    for name in ("cc", "cs", "c1", "cmu", "damps", "chiN", "mueff", "mu"):
        assert getattr(blocked, name) == getattr(member, name)
    bstate = blocked.init(ex.PRNGKey(2))
    mstates = [member.init(ex.PRNGKey(20 + s)) for s in range(B)]
    target = torch.randn(B * d, generator=torch.Generator().manual_seed(3))
    for gen in range(15):
        pop, bstate = blocked.ask(bstate)
        assert pop.shape == (P, B * d)
        fitness = ((pop - target) ** 2).sum(1) + 0.1 * torch.sin(5 * pop).sum(1)
        bstate = blocked.tell(bstate, fitness)
        for s in range(B):
            _, ms = member.ask(mstates[s])
            mstates[s] = ms = member.tell(ms.replace(population=pop.reshape(P, B, d)[:, s].clone()), fitness)
            for name in ("mean", "ps", "pc", "C"):
                torch.testing.assert_close(getattr(bstate, name).reshape(B, d)[s], getattr(ms, name), rtol=1e-4, atol=1e-6,
                                           msg=f"{name} gen {gen}")
            torch.testing.assert_close(bstate.sigma[s], ms.sigma, rtol=1e-4, atol=1e-7)
    assert bstate.sigma.shape == (B,) and len({round(float(v), 6) for v in bstate.sigma}) == B


@pytest.mark.parametrize("name", ["DE", "JaDE", "SHADE", "SepCMAES"])
def test_blocked_members_with_a_fitness_per_block_follow_the_member_update(name):
    """Co-evolution form: every block is told a fitness column of its own.  Fed the instance's candidates block by block
    and that column, every member's own tell() reproduces the block (population, fitness, best index, memories; mean,
    step size, paths and variances for SepCMAES)."""
    B, d, P = 3, 4, 24
    lb, ub = torch.linspace(-3, -1, d), torch.linspace(1, 3, d)
    member = {"DE": lambda: algorithms.DE(lb, ub, P, batch_size=8), "JaDE": lambda: algorithms.JaDE(lb, ub, P),
              "SHADE": lambda: algorithms.SHADE(lb, ub, P, memory_size=5),
              "SepCMAES": lambda: algorithms.SepCMAES(torch.linspace(-1, 1, d), 0.8, pop_size=P)}[name]()
    blocked = member.blocked(B, shared_fitness=False)
    assert blocked.block_fitness and not member.block_fitness
    bstate = blocked.init(ex.PRNGKey(5))
    mstates = [member.init(ex.PRNGKey(50 + s)) for s in range(B)]
    if name != "SepCMAES":
        assert bstate.fitness.shape == (P, B) and bstate.best_index.shape == (B,)
        mstates = [m.replace(population=bstate.population.reshape(P, B, d)[:, s].clone()) for s, m in enumerate(mstates)]
    target = torch.randn(B, d, generator=torch.Generator().manual_seed(1)) * 0.5
    for gen in range(10):
        cand, bstate = blocked.ask(bstate)
        rows = cand.shape[0]
        fit = ((cand.reshape(rows, B, d) - target) ** 2).sum(-1) + 0.2 * torch.sin(7 * cand.reshape(rows, B, d)).sum(-1)   # (rows, B)
        after = blocked.tell(bstate, fit)
        for s in range(B):
            if name == "SepCMAES":
                _, ms = member.ask(mstates[s])
                ms = member.tell(ms.replace(population=cand.reshape(P, B, d)[:, s].clone()), fit[:, s].contiguous())
                for field in ("mean", "ps", "pc", "C"):
                    torch.testing.assert_close(getattr(after, field).reshape(B, d)[s], getattr(ms, field), rtol=1e-4, atol=1e-6)
                torch.testing.assert_close(after.sigma[s], ms.sigma, rtol=1e-4, atol=1e-7)
            else:
                ms = mstates[s].replace(trial_vectors=cand.reshape(rows, B, d)[:, s].clone(), start_index=bstate.start_index,
                                        start_index_dev=bstate.start_index_dev.clone())
                if name in ("JaDE", "SHADE"):
                    ms = ms.replace(F_vect=bstate.F_vect.reshape(P, B)[:, s].clone(), CR_vect=bstate.CR_vect.reshape(P, B)[:, s].clone())
                ms = member.tell(ms, fit[:, s].contiguous())
                assert torch.equal(after.population.reshape(P, B, d)[:, s], ms.population), (gen, s)
                assert torch.equal(after.fitness[:, s], ms.fitness) and int(after.best_index[s]) == int(ms.best_index)
                if name == "JaDE":
                    torch.testing.assert_close(after.F_u[s], ms.F_u, rtol=1e-5, atol=1e-6)
                    torch.testing.assert_close(after.CR_u[s], ms.CR_u, rtol=1e-5, atol=1e-6)
                if name == "SHADE":
                    torch.testing.assert_close(after.Memory_FCR[:, s], ms.Memory_FCR, rtol=1e-5, atol=1e-6)
            mstates[s] = ms
        bstate = after
    if name != "SepCMAES":
        assert len({int(v) for v in bstate.best_index}) > 1 or B == 1        # the blocks rank the individuals differently


@pytest.mark.parametrize("name", ["SHADE", "SepCMAES"])
def test_vectorized_coevolution_runs_identical_members_as_one_blocked_instance(name):
    results = {}
    for fuse in (False, "auto"):
        lb, ub = torch.full((5,), -10.0), torch.full((5,), 10.0)
        base = algorithms.SHADE(lb, ub, 40) if name == "SHADE" else algorithms.SepCMAES(torch.full((5,), 4.0), 3.0, pop_size=40)
        algo = algorithms.VectorizedCoevolution(base, dim=20, num_subpops=4, fuse=fuse)
        assert (algo.batched is not None) == (fuse == "auto") and len(algo.base_algorithms) == (1 if fuse == "auto" else 4)
        results[fuse], _ = _run(algo, 120, problems.numerical.Sphere())
    assert results["auto"] < 0.5 and results[False] < 0.5


def test_blocked_de_draws_parents_per_block():
    """DE/rand/1 with CR = 1: every block of every trial row is x[r1] + F (x[r2] - x[r3]) of THAT block's sub-vectors,
    with r1, r2, r3 distinct and different from the row, and the blocks of a row use different parents."""
    import itertools

    B, d, P, F = 4, 3, 7, 0.5
    member = algorithms.DE(torch.full((d,), -100.0), torch.full((d,), 100.0), P, differential_weight=F, cross_probability=1.0,
                           batch_size=P, mean=0.0, stdvar=1.0)              # far inside the bounds: nothing is clipped
    blocked = member.blocked(B)
    state = blocked.init(ex.PRNGKey(11))
    trial, _ = blocked.ask(state)
    pop = state.population.reshape(P, B, d)
    found = {}
    for i in range(P):
        for s in range(B):
            hits = [(a, b, c) for a, b, c in itertools.permutations([j for j in range(P) if j != i], 3)
                    if torch.allclose(pop[a, s] + F * (pop[b, s] - pop[c, s]), trial.reshape(P, B, d)[i, s], rtol=1e-5, atol=1e-4)]
            assert len(hits) == 1, (i, s, hits)
            found[i, s] = hits[0]
    assert any(len({found[i, s] for s in range(B)}) > 1 for i in range(P))


@pytest.mark.parametrize("name", ["DE", "SHADE", "SepCMAES"])
def test_clustered_de_family_runs_as_one_blocked_instance(name):
    results = {}
    for fuse in (False, "auto"):
        lb, ub = torch.full((5,), -10.0), torch.full((5,), 10.0)
        base = {"DE": lambda: algorithms.DE(lb, ub, 40), "SHADE": lambda: algorithms.SHADE(lb, ub, 40),
                "SepCMAES": lambda: algorithms.SepCMAES(torch.full((5,), 4.0), 3.0, pop_size=40)}[name]()
        algo = algorithms.ClusterdAlgorithm(base_algorithm=base, dim=20, num_cluster=4, fuse=fuse)
        assert len(algo.members) == (1 if fuse == "auto" else 4) and algo.fused == (fuse == "auto")
        results[fuse], _ = _run(algo, 150, problems.numerical.Sphere())
    assert results["auto"] < 1.0 and results[False] < 1.0


def test_clustered_cma_es_runs_as_one_batched_instance():
    results = {}
    for fuse in (False, "auto"):
        base = algorithms.CMAES(torch.full((10,), -10.0), init_stdev=6, pop_size=10)
        algo = algorithms.ClusterdAlgorithm(base_algorithm=base, dim=40, num_cluster=4, fuse=fuse)
        assert len(algo.members) == (1 if fuse == "auto" else 4)
        results[fuse], state = _run(algo, 300)
    assert results["auto"] < 3 and results[False] < 3


def test_vectorized_coevolution_batches_identical_cma_members():
    results = {}
    for fuse in (False, "auto"):
        base = algorithms.CMAES(torch.full((10,), 5.0), init_stdev=3.0, pop_size=16)
        algo = algorithms.VectorizedCoevolution(Stateful.stack([base] * 4), dim=40, num_subpops=4, random_subpop=True, fuse=fuse)
        assert (algo.batched is not None) == (fuse == "auto")
        results[fuse], _ = _run(algo, 150)
    assert results["auto"] < 1.0 and results[False] < 1.0


@pytest.mark.parametrize("random_subpop", [True, False])
def test_vectorized_coevolution(random_subpop):
    base = algorithms.CSO(lb=torch.full((20,), -32.0), ub=torch.full((20,), 32.0), pop_size=100)
    algo = algorithms.VectorizedCoevolution(base_algorithms=Stateful.stack([base] * 2), dim=40, num_subpops=2,
                                            random_subpop=random_subpop)
    best, _ = _run(algo, 200)
    assert best < 0.5


@pytest.mark.parametrize("random_subpop", [True, False])
def test_coevolution(random_subpop):
    base = algorithms.CSO(lb=torch.full((20,), -32.0), ub=torch.full((20,), 32.0), pop_size=100)
    algo = algorithms.Coevolution(Stateful.stack([base] * 2), dim=40, num_subpops=2, random_subpop=random_subpop)
    best, _ = _run(algo, 400)
    assert best < 0.5


def test_random_mask_cso():
    base = algorithms.CSO(lb=torch.full((10,), -32.0), ub=torch.full((10,), 32.0), pop_size=50)
    algo = algorithms.RandomMaskAlgorithm(base_algorithm=base, dim=40, num_cluster=4, num_mask=2, change_every=10, pop_size=50)
    first, _ = _run(algo, 2)
    best, state = _run(algo, 150)
    assert best < first and best < 12
    assert len(state.get_child_state("algorithm").active) == 2


def test_tree_algorithm():
    params = {"w": torch.zeros(3, 2), "b": torch.zeros(2)}
    lb = {"w": -torch.ones(6), "b": -torch.ones(2)}
    ub = {"w": torch.ones(6), "b": torch.ones(2)}
    algo = algorithms.TreeAlgorithm(lambda lo, hi: algorithms.PSO(lb=lo, ub=hi, pop_size=20), params, lb, ub)

    class Quadratic(ex.Problem):
        def evaluate(self, state, pop):
            return (pop["w"] ** 2).sum((1, 2)) + ((pop["b"] - 0.5) ** 2).sum(1), state

    monitor = EvalMonitor()
    wf = workflows.StdWorkflow(algo, Quadratic(), monitors=[monitor])
    state = wf.init(ex.PRNGKey(0))
    pop, _ = wf.algorithm.init_ask(state.get_child_state("algorithm")) if False else (None, None)
    for _ in range(60):
        state = wf.step(state)
    assert float(monitor.get_best_fitness()) < 0.05


def test_module_stack_gives_members_distinct_states():
    base = algorithms.PSO(lb=torch.zeros(2), ub=torch.ones(2), pop_size=4)
    stack = Stateful.stack([base, base, base])
    assert len(stack) == 3 and stack[0] is not stack[1]
    state = stack.init(ex.PRNGKey(0))
    pops = [state.get_child_state(n).population for n in ("m0000", "m0001", "m0002")]
    assert not torch.equal(pops[0], pops[1]) and not torch.equal(pops[1], pops[2])


@pytest.mark.gpu
def test_batched_clustered_cma_es_on_gpu():
    """100 blocks of 10 variables as one batched CMA-ES on the GPU: improves on the sphere (every block sees the noise
    of the other 99 in its fitness, so progress is slow by construction), state stays on the device."""
    base = algorithms.CMAES(torch.full((10,), 3.0, device="cuda"), init_stdev=2.0, pop_size=32)
    algo = algorithms.ClusterdAlgorithm(base, dim=1000, num_cluster=100)
    assert len(algo.members) == 1
    monitor = EvalMonitor()
    wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), monitors=[monitor])
    state = wf.init(ex.PRNGKey(1).cuda())
    first = None
    for g in range(150):
        state = wf.step(state)
        if g == 0:
            first = float(monitor.get_best_fitness())
    inner = state.get_child_state("algorithm").get_child_state("members").get_child_state("m0000")
    assert inner.C.shape == (100, 10, 10) and inner.C.is_cuda and inner.sigma.shape == (100,)
    assert float(monitor.get_best_fitness()) < 0.7 * first


@pytest.mark.gpu
@pytest.mark.parametrize("base", ["DE", "SHADE", "SepCMAES"])
def test_member_loop_container_records_with_its_members_host_counters(base):
    """A container over members that keep host-side counters (DE's sliding window offset, CMA-ES's generation count)
    hands those counters — with their paths below its own state — to the recording, so the whole member loop replays
    from one HIP graph and equals eager issue."""
    dev = torch.device("cuda")
    lb, ub = torch.full((5,), -10.0, device=dev), torch.full((5,), 10.0, device=dev)
    make = {"DE": lambda: algorithms.DE(lb, ub, 64), "SHADE": lambda: algorithms.SHADE(lb, ub, 64),
            "SepCMAES": lambda: algorithms.SepCMAES(torch.full((5,), 3.0, device=dev), 1.0, pop_size=64)}[base]
    out = {}
    for mode in (False, True):
        algo = algorithms.ClusterdAlgorithm(make(), dim=40, num_cluster=8, fuse=False)
        assert not algo.fused and len(algo.members) == 8
        wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), graph_step=mode)
        state = wf.init(ex.PRNGKey(7).to(dev))
        for _ in range(25):
            state = wf.step(state)
        torch.cuda.synchronize()
        m0 = state.get_child_state("algorithm").get_child_state("members").get_child_state("m0003")
        out[mode] = (m0.mean if base == "SepCMAES" else m0.population).clone()
        if mode:
            assert wf.graph_replays >= 15
    assert torch.equal(out[True], out[False])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["DE_best2", "SHADE"])
def test_blocked_de_trial_kernel_with_a_fitness_per_block(name, monkeypatch):
    """Blocks with a fitness of their own: the fused launch reads the best individual of every block and a p-best set
    per block, like the tensor formulation."""
    from evox_amd.ops import de as de_ops

    dev = torch.device("cuda")
    B, d, P = 7, 10, 300
    lb, ub = torch.linspace(-5, -4, d, device=dev), torch.linspace(4, 5, d, device=dev)
    member = algorithms.SHADE(lb, ub, P) if name == "SHADE" else algorithms.DE(lb, ub, P, base_vector="best",
                                                                             num_difference_vectors=2)
    blocked = member.blocked(B, shared_fitness=False)
    state = blocked.init(ex.PRNGKey(4).to(dev))
    fit = torch.rand(P, B, device=dev)
    state = state.replace(fitness=fit, best_index=torch.argmin(fit, dim=0))
    assert len({int(v) for v in state.best_index}) > 1
    fused, _ = blocked.ask(state)
    monkeypatch.setattr(de_ops, "supported", lambda *a, **k: False)
    plain, _ = blocked.ask(state)
    torch.testing.assert_close(fused, plain, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SHADE", "SepCMAES", "CMAES"])
def test_vectorized_coevolution_blocked_replays_from_a_graph(name):
    """The one instance that stands for the members is a child of the container once: its host counters (generation
    counts) are found by the recording, and the replayed generations equal eager issue."""
    dev = torch.device("cuda")
    lb, ub = torch.full((10,), -10.0, device=dev), torch.full((10,), 10.0, device=dev)
    out = {}
    for mode in (False, True):
        base = {"SHADE": lambda: algorithms.SHADE(lb, ub, 128),
                "SepCMAES": lambda: algorithms.SepCMAES(torch.full((10,), 3.0, device=dev), 2.0, pop_size=128),
                "CMAES": lambda: algorithms.CMAES(torch.full((10,), 3.0, device=dev), 2.0, pop_size=128)}[name]()
        algo = algorithms.VectorizedCoevolution(base, dim=200, num_subpops=20)
        assert algo.batched is not None
        wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), graph_step=mode)
        state = wf.init(ex.PRNGKey(7).to(dev))
        for _ in range(40):
            state = wf.step(state)
        torch.cuda.synchronize()
        out[mode] = state.get_child_state("algorithm").best_dec.clone()
        if mode:
            assert wf.graph_replays >= 20
    if name == "CMAES":                  # batched library eigendecompositions differ in the last bits between the runs
        torch.testing.assert_close(out[True], out[False], rtol=1e-3, atol=1e-3)
    else:
        assert torch.equal(out[True], out[False])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["DE", "DE_best2", "JaDE", "SHADE", "CoDE"])
def test_blocked_de_trial_kernel_matches_tensor_formulation(name, monkeypatch):
    """The fused variation launch over virtual rows (individual, block) draws the same random elements as the tensor
    formulation of the blocked instance."""
    from evox_amd.ops import de as de_ops

    dev = torch.device("cuda")
    B, d, P = 7, 10, 300
    lb, ub = torch.linspace(-5, -4, d, device=dev), torch.linspace(4, 5, d, device=dev)
    member = {"DE": lambda: algorithms.DE(lb, ub, P, batch_size=64), "JaDE": lambda: algorithms.JaDE(lb, ub, P),
              "DE_best2": lambda: algorithms.DE(lb, ub, P, base_vector="best", num_difference_vectors=2),
              "CoDE": lambda: algorithms.CoDE(lb, ub, P), "SHADE": lambda: algorithms.SHADE(lb, ub, P)}[name]()
    blocked = member.blocked(B)
    state = blocked.init(ex.PRNGKey(4).to(dev))
    state = state.replace(fitness=torch.rand(P, device=dev), best_index=torch.tensor(17, device=dev))
    fused, _ = blocked.ask(state)
    monkeypatch.setattr(de_ops, "supported", lambda *a, **k: False)
    plain, _ = blocked.ask(state)
    assert fused.shape == plain.shape == ({"DE": 64, "DE_best2": 100, "CoDE": 3 * P}.get(name, P), B * d)      # DE varies a window
    torch.testing.assert_close(fused, plain, rtol=1e-5, atol=1e-5)
    assert bool((fused >= blocked.lb).all() and (fused <= blocked.ub).all())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["DE", "SHADE", "SepCMAES"])
def test_blocked_container_replays_from_a_graph(name):
    """ClusterdAlgorithm over a DE-family base is one blocked instance; its recorded generation equals eager issue."""
    dev = torch.device("cuda")
    lb, ub = torch.full((10,), -10.0, device=dev), torch.full((10,), 10.0, device=dev)
    out = {}
    for mode in (False, True):
        base = {"DE": lambda: algorithms.DE(lb, ub, 256), "SHADE": lambda: algorithms.SHADE(lb, ub, 256),
                "SepCMAES": lambda: algorithms.SepCMAES(torch.full((10,), 3.0, device=dev), 2.0, pop_size=256)}[name]()
        algo = algorithms.ClusterdAlgorithm(base, dim=1000, num_cluster=100)
        assert algo.fused and len(algo.members) == 1
        monitor = EvalMonitor()
        wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), monitors=[monitor], graph_step=mode)
        state = wf.init(ex.PRNGKey(7).to(dev))
        for _ in range(40):
            state = wf.step(state)
        torch.cuda.synchronize()
        inner = state.get_child_state("algorithm").get_child_state("members").get_child_state("m0000")
        out[mode] = inner.population.clone()
        if mode:
            assert wf.graph_replays >= 25
        assert inner.population.shape == (256, 1000)
    assert torch.equal(out[True], out[False])
