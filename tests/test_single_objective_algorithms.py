"""Convergence of the single-objective algorithms on a 5-D Sphere through StdWorkflow (the
reference's integration test strategy, tests/test_single_objective_algorithms.py:24-167, with the
same thresholds: best fitness < 0.1, < 1 for OpenES)."""
import warnings

import pytest
import torch

import evox_amd as ex
from evox_amd import algorithms as A
from evox_amd import monitors, problems, workflows

DIM = 5
LB, UB = torch.full((DIM,), -10.0), torch.full((DIM,), 10.0)
C = torch.full((DIM,), 3.0)


def run(algo, steps, problem=None, seed=42):
    mon = monitors.EvalMonitor()
    wf = workflows.StdWorkflow(algo, problem or problems.numerical.Sphere(), monitors=[mon])
    state = wf.init(ex.PRNGKey(seed))
    for _ in range(steps):
        state = wf.step(state)
    return float(mon.get_best_fitness()), state


CASES = {
    "PSO": (lambda: A.PSO(LB, UB, 100), 100, 0.1),
    "CSO": (lambda: A.CSO(lb=LB, ub=UB, pop_size=100), 200, 0.1),
    "CLPSO": (lambda: A.CLPSO(LB, UB, 100, 0.5, 1.5, torch.full((100,), 0.2)), 200, 0.1),
    "SLPSOGS": (lambda: A.SLPSOGS(LB, UB, 100, 0.2, 0.7), 200, 0.1),
    "SLPSOUS": (lambda: A.SLPSOUS(LB, UB, 100, 0.2, 0.7), 200, 0.1),
    "DMSPSOEL": (lambda: A.DMSPSOEL(LB, UB, 10, 8, 20, 5, 300, 0.7, 1.5, 1.5, 1.0, 1.0), 300, 0.1),
    "FIPS": (lambda: A.FIPS(LB, UB, 100), 300, 0.1),
    "SwmmPSO": (lambda: A.SwmmPSO(LB, UB, 100, shortcut=5), 200, 0.1),
    "FSPSO": (lambda: A.FSPSO(LB, UB, 100), 200, 0.1),
    "DE": (lambda: A.DE(LB, UB, 100, batch_size=100), 100, 0.1),
    "DE-best-window": (lambda: A.DE(LB, UB, 100, base_vector="best", batch_size=50, num_difference_vectors=2), 200, 0.1),
    "ODE": (lambda: A.ODE(LB, UB, 100), 200, 0.1),
    "JaDE": (lambda: A.JaDE(LB, UB, 100), 100, 0.1),
    "SHADE": (lambda: A.SHADE(LB, UB, 100), 100, 0.1),
    "SaDE": (lambda: A.SaDE(LB, UB, 100), 100, 0.1),
    "CoDE": (lambda: A.CoDE(LB, UB, 100), 100, 0.1),
    "CMAES": (lambda: A.CMAES(C, 1.0, pop_size=20), 100, 0.1),
    "CMAES-newton-schulz": (lambda: A.CMAES(C, 1.0, pop_size=20, decomposition="newton_schulz"), 100, 0.1),
    "SepCMAES": (lambda: A.SepCMAES(C, 1.0, pop_size=20), 100, 0.1),
    "IPOPCMAES": (lambda: A.IPOPCMAES(C, 1.0, pop_size=20), 100, 0.1),
    "BIPOPCMAES": (lambda: A.BIPOPCMAES(C, 1.0, pop_size=20), 100, 0.1),
    "MAES": (lambda: A.MAES(C, 1.0, pop_size=20), 200, 0.1),
    "LMMAES": (lambda: A.LMMAES(C, 1.0, pop_size=20), 300, 0.5),
    "AMaLGaM": (lambda: A.AMaLGaM(C, 1.0, pop_size=20), 100, 0.1),
    "IndependentAMaLGaM": (lambda: A.IndependentAMaLGaM(C, 1.0, pop_size=20), 100, 0.1),
    "RMES": (lambda: A.RMES(C, 1.0, pop_size=20), 100, 0.1),
    "CR_FM_NES": (lambda: A.CR_FM_NES(20, torch.full((10,), 1.0), sigma=1.0), 300, 0.1),
    "OpenES": (lambda: A.OpenES(C, 100, 0.05, 0.1), 500, 1.0),
    "OpenES-adam": (lambda: A.OpenES(C, 100, 0.05, 0.1, optimizer="adam"), 500, 1.0),
    "PGPE-adam": (lambda: A.PGPE(100, C, "adam", center_learning_rate=0.3, stdev_init=1.0), 300, 0.1),
    "PGPE-clipup": (lambda: A.PGPE(100, C, "clipup", center_learning_rate=0.3, stdev_init=1.0), 300, 0.1),
    "ARS": (lambda: A.ARS(100, C, lr=0.1, sigma=0.1), 400, 0.1),
    "DES": (lambda: A.DES(100, C, sigma_init=1.0), 200, 0.1),
    "ESMC": (lambda: A.ESMC(101, C, lr=0.1, sigma=0.1), 400, 0.1),
    "SNES": (lambda: A.SNES(100, C), 200, 0.1),
    "SeparableNES": (lambda: A.SeparableNES(C, torch.ones(DIM), 100), 200, 0.1),
    "XNES": (lambda: A.XNES(C, torch.eye(DIM), 100), 200, 0.1),
    "GuidedES": (lambda: A.GuidedES(100, C, lrate_init=0.5, sigma_init=0.1), 400, 0.1),
    "Noise_reuse_es": (lambda: A.Noise_reuse_es(100, C, lr=0.1, sigma=0.1), 400, 0.1),
    "PersistentES": (lambda: A.PersistentES(100, C, lr=0.1, sigma=0.1, T=10, K=10), 400, 0.1),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_converges_on_sphere(name):
    make, steps, threshold = CASES[name]
    best, _ = run(make(), steps)
    assert best < threshold, f"{name}: best fitness {best}"


def test_asebo_improves_in_twenty_dimensions():
    c = torch.full((20,), 1.0)
    best, _ = run(A.ASEBO(20, c, lr=0.05, sigma=0.1, subspace_dims=15), 600)
    assert best < 0.5 * 20.0


def test_les_runs_with_supplied_parameters():
    params = A.init_les_params(ex.PRNGKey(0))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        best, state = run(A.LES(16, C, sigma_init=1.0, net_params=params), 20)
    alg = state.get_child_state("algorithm")
    assert torch.isfinite(alg.mean).all() and torch.isfinite(alg.sigma).all() and alg.gen_counter == 20


def test_ipop_restart_grows_population():
    algo = A.IPOPCMAES(C, 1.0, pop_size=8, stagnation_threshold=3)

    class Flat(ex.Problem):
        def evaluate(self, state, x):
            return torch.ones(x.shape[0]), state

    wf = workflows.StdWorkflow(algo, Flat())
    state = wf.init(ex.PRNGKey(0))
    for _ in range(12):
        state = wf.step(state)
    alg = state.get_child_state("algorithm")
    assert alg.restarts >= 1 and alg.pop_size == 8 * 2 ** alg.restarts
    assert alg.population.shape[0] == alg.pop_size


def test_same_key_same_trajectory():
    a, _ = run(A.SHADE(LB, UB, 50), 30, seed=7)
    b, _ = run(A.SHADE(LB, UB, 50), 30, seed=7)
    c, _ = run(A.SHADE(LB, UB, 50), 30, seed=8)
    assert a == b and a != c


def test_opt_direction_max():
    class NegSphere(ex.Problem):
        def evaluate(self, state, x):
            return -(x * x).sum(1), state

    mon = monitors.EvalMonitor()
    wf = workflows.StdWorkflow(A.PSO(LB, UB, 50), NegSphere(), monitors=[mon], opt_direction="max")
    state = wf.init(ex.PRNGKey(1))
    for _ in range(100):
        state = wf.step(state)
    assert -0.1 < float(mon.get_best_fitness()) <= 0.0


def test_asebo_population_larger_than_gradient_history_stays_finite():
    """With more directions requested than the gradient history holds, the orthogonal complement is empty and
    the subspace weight is 0/0; sampling must fall back to the isotropic distribution instead of failing."""
    algo = A.ASEBO(64, torch.full((12,), 3.0), lr=0.05, sigma=0.1, subspace_dims=6)
    wf = workflows.StdWorkflow(algo, problems.numerical.Rastrigin())
    state = wf.init(ex.PRNGKey(5))
    for _ in range(12):
        state = wf.step(state)
    a = state.get_child_state("algorithm")
    assert torch.isfinite(a.center).all() and torch.isfinite(a.alpha)


@pytest.mark.parametrize("name", ["DE", "ODE", "CoDE", "SaDE", "JaDE", "SHADE", "PSO"])
def test_migrate_hook_replaces_the_worst_residents(name):
    """Every DE variant and PSO take immigrants (``IslandWorkflow`` calls ``algorithm.migrate``): an immigrant that
    improves on one of the worst residents takes its place, one that does not is dropped."""
    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows

    lb, ub = torch.full((6,), -5.0), torch.full((6,), 5.0)
    algo = getattr(algorithms, name)(lb, ub, 40) if name != "PSO" else algorithms.PSO(lb=lb, ub=ub, pop_size=40)
    wf = workflows.StdWorkflow(algo, problems.numerical.Sphere())
    state = wf.init(ex.PRNGKey(2))
    for _ in range(3):
        state = wf.step(state)
    a = state.get_child_state("algorithm")
    fit_name = "local_best_fitness" if name == "PSO" else "fitness"
    before = getattr(a, fit_name).clone()
    foreign = torch.zeros(3, 6)
    foreign[2] = 100.0                                       # two perfect immigrants and a hopeless one
    ffit = (foreign ** 2).sum(1)
    b = algo.migrate(a, foreign, ffit)
    after = getattr(b, fit_name)
    assert int((after == 0).sum()) == 2 and float(after.max()) <= float(before.max())
    assert float(after.sum()) < float(before.sum())
    if name == "PSO":
        assert float(b.global_best_fitness) == 0.0 and float(b.global_best_location.abs().max()) == 0.0
    else:
        assert float(b.fitness[b.best_index]) == 0.0


def test_newton_schulz_plan_picks_the_driver_by_conditioning():
    """Host decisions of a factorisation (``NewtonSchulzSqrt.plan``) from the condition bound two calls ago: mirrored
    plane products while it is small, plain plane products with two and then three planes beyond, the fp32 library
    products only past ``PLANE_COND_MAX``; the choice is stored in the control (``ns_sym``) for recorded steps."""
    from evox_amd.ops.cma import NewtonSchulzSqrt

    ns = NewtonSchulzSqrt(planes=2, initial_cond=1.0)
    ctrl = ns.init_control(8, torch.device("cpu"))

    def plan_for(cond):
        rec = torch.tensor([cond / 2.0, cond, 1.0, 1.0, 1.0, 1e-5, 0.0])
        c = dict(ctrl, ns_calls=5, ns_used1=3, ns_used2=3, ns_res1=rec, ns_res2=rec)
        pl = ns.plan(c, False, True, True)
        return pl, ns.commit(c, pl, rec, c["ns_vec"], False)["ns_sym"]

    pl, sym = plan_for(4.0)
    assert pl["symmetric"] and not pl["full"] and pl["planes"] == 2 and sym == 1
    pl, sym = plan_for(500.0)
    assert pl["symmetric"] and pl["full"] and pl["planes"] == 2 and sym == 2
    pl, sym = plan_for(8000.0)
    assert pl["symmetric"] and pl["full"] and pl["planes"] == 3 and sym == 3
    pl, sym = plan_for(1e6)
    assert not pl["symmetric"] and not pl["full"] and sym == 0
    # a recorded step keeps the driver of the call before the recording
    for stored, want in ((1, (True, False, 2)), (2, (True, True, 2)), (3, (True, True, 3)), (0, (False, False, 2))):
        pl = ns.plan(dict(ctrl, ns_calls=5, ns_sym=stored), True, True, True)
        assert (pl["symmetric"], pl["full"], pl["planes"]) == want


def test_sade_memory_push_matches_the_sort_formulation():
    """The values of a column's selected rows, newest (highest row) first, NaN-padded: the cumulative-sum scatter against
    the stable-sort formulation it replaced."""
    from evox_amd.algorithms.so.de_variants.adaptive import _newest_first

    g = torch.Generator().manual_seed(0)
    for rows, depth in ((100, 50), (30, 50), (1024, 50), (7, 3)):
        for _ in range(5):
            sel = torch.rand(rows, 4, generator=g) < torch.rand(1, generator=g)
            values = torch.rand(rows, generator=g)
            order = torch.argsort(sel.to(torch.int64) * (torch.arange(rows)[:, None] + 1), dim=0, descending=True, stable=True)
            want = torch.where(torch.gather(sel, 0, order), values[order], torch.full((), float("nan")))[:depth]
            if want.shape[0] < depth:
                want = torch.cat([want, torch.full((depth - want.shape[0], 4), float("nan"))], dim=0)
            got = _newest_first(sel, values, depth)
            assert torch.equal(torch.isnan(got), torch.isnan(want))
            assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
