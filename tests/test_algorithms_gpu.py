"""Every algorithm takes a few generations with all of its state on the GPU (device placement,
kernels, no accidental host tensors)."""
import pytest
import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems, workflows
from evox_amd.core.tree import tree_leaves

import test_multi_objective_algorithms as mo_cases
import test_single_objective_algorithms as so_cases

pytestmark = pytest.mark.gpu


@pytest.fixture()
def so_on_gpu(cuda, monkeypatch):
    monkeypatch.setattr(so_cases, "LB", so_cases.LB.to(cuda))
    monkeypatch.setattr(so_cases, "UB", so_cases.UB.to(cuda))
    monkeypatch.setattr(so_cases, "C", so_cases.C.to(cuda))
    return cuda


@pytest.mark.parametrize("name", sorted(so_cases.CASES))
def test_single_objective_on_gpu(so_on_gpu, name):
    make, _, _ = so_cases.CASES[name]
    if name == "CR_FM_NES":
        algo = algorithms.CR_FM_NES(20, torch.full((10,), 1.0, device=so_on_gpu), sigma=1.0)
    elif name == "SeparableNES":
        algo = algorithms.SeparableNES(so_cases.C, torch.ones(so_cases.DIM, device=so_on_gpu), 100)
    elif name == "XNES":
        algo = algorithms.XNES(so_cases.C, torch.eye(so_cases.DIM, device=so_on_gpu), 100)
    else:
        algo = make()
    mon = monitors.EvalMonitor()
    wf = workflows.StdWorkflow(algo, problems.numerical.Sphere(), monitors=[mon])
    state = wf.init(ex.PRNGKey(1))
    first = None
    for i in range(12):
        state = wf.step(state)
        if i == 0:
            first = float(mon.get_best_fitness())
    best = float(mon.get_best_fitness())
    assert best == best and best <= first
    tensors = [t for t in tree_leaves(state.get_child_state("algorithm")) if isinstance(t, torch.Tensor)]
    assert tensors and all(t.is_cuda for t in tensors), [t.device for t in tensors if not t.is_cuda]


@pytest.mark.parametrize("name,kw,steps,bound", mo_cases.CASES, ids=[c[0] for c in mo_cases.CASES])
def test_multi_objective_on_gpu(cuda, name, kw, steps, bound):
    n, m = mo_cases.N, mo_cases.M
    algo = getattr(algorithms, name)(lb=torch.zeros(n, device=cuda), ub=torch.ones(n, device=cuda), n_objs=m, pop_size=mo_cases.POP,
                                     **kw)
    wf = workflows.StdWorkflow(algo, problems.numerical.DTLZ2(m=m))
    state = wf.init(ex.PRNGKey(2))
    for _ in range(8):
        state = wf.step(state)
    alg = state.get_child_state("algorithm")
    assert alg.fitness.is_cuda and alg.population.is_cuda
    assert torch.isfinite(alg.fitness[torch.isfinite(alg.fitness).all(1)]).all()
