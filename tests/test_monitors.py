"""Monitor behaviour; scenarios follow the reference's ``tests/test_monitors.py:8-140``."""
import pytest
import torch

import evox_amd as ex
from evox_amd import algorithms, problems, workflows
from evox_amd.monitors import EvalMonitor, PopMonitor, StdMOMonitor, StdSOMonitor


def test_std_so_monitor():
    monitor = StdSOMonitor(record_topk=1, record_fit_history=True)
    pop1 = torch.arange(15).reshape(3, 5)
    fit1 = torch.arange(3)
    monitor.record_pop(pop1)
    monitor.record_fit(fit1)
    assert monitor.get_best_fitness() == 0
    assert monitor.get_topk_fitness() == 0
    assert (monitor.get_best_solution() == pop1[0]).all()
    assert (monitor.get_topk_solutions() == pop1[0:1]).all()
    pop2, fit2 = -pop1, -fit1
    monitor.record_pop(pop2)
    monitor.record_fit(fit2)
    assert monitor.get_best_fitness() == -2
    assert (monitor.get_best_solution() == pop2[2]).all()
    assert (monitor.get_topk_solutions() == pop2[2:3]).all()


def test_std_mo_monitor():
    monitor = StdMOMonitor(record_pf=True, record_fit_history=True)
    pop1 = torch.arange(15).reshape(3, 5)
    monitor.record_pop(pop1)
    monitor.record_fit(torch.tensor([[1.0, 2], [3, 1], [5, 6]]))
    assert (monitor.get_pf_fitness() == torch.tensor([[1.0, 2], [3, 1]])).all()
    assert (monitor.get_pf_solutions() == pop1[:2]).all()
    pop2 = -pop1
    monitor.record_pop(pop2)
    monitor.record_fit(torch.tensor([[0.5, 1.5], [7, 8], [0, 10]]))
    assert (monitor.get_pf_fitness() == torch.tensor([[3, 1], [0.5, 1.5], [0, 10]])).all()
    assert (monitor.get_pf_solutions() == torch.cat([pop1[1:2], pop2[[0, 2]]])).all()


@pytest.mark.parametrize("full_fit_history,full_sol_history,topk",
                         [(False, False, 1), (False, False, 2), (False, True, 1), (False, True, 2), (True, False, 1),
                          (True, False, 2), (True, True, 1), (True, True, 2)])
def test_eval_monitor_with_so(full_fit_history, full_sol_history, topk):
    monitor = EvalMonitor(full_fit_history=full_fit_history, full_sol_history=full_sol_history, topk=topk)
    pop1 = torch.arange(15).reshape(3, 5)
    fit1 = torch.arange(3)
    monitor.post_eval(None, pop1, None, fit1)
    assert monitor.get_best_fitness() == 0
    assert (monitor.get_topk_fitness() == fit1[:topk]).all()
    assert (monitor.get_best_solution() == pop1[0]).all()
    assert (monitor.get_topk_solutions() == pop1[:topk]).all()
    pop2, fit2 = -pop1, -fit1
    monitor.post_eval(None, pop2, None, fit2)
    assert monitor.get_best_fitness() == -2
    assert (monitor.get_topk_fitness() == fit2[-topk:].flip(0)).all()
    assert (monitor.get_best_solution() == pop2[-1]).all()
    assert (monitor.get_topk_solutions() == pop2[-topk:].flip(0)).all()


@pytest.mark.parametrize("full_fit_history,full_sol_history", [(False, False), (False, True), (True, False), (True, True)])
def test_eval_monitor_with_mo(full_fit_history, full_sol_history):
    monitor = EvalMonitor(full_fit_history=full_fit_history, full_sol_history=full_sol_history)
    pop1 = torch.arange(15).reshape(3, 5)
    fit1 = torch.arange(6).reshape(3, 2)
    monitor.post_eval(None, pop1, None, fit1)
    assert (monitor.get_latest_fitness() == fit1).all()
    assert (monitor.get_latest_solution() == pop1).all()
    monitor.post_eval(None, -pop1, None, -fit1)
    assert (monitor.get_latest_fitness() == -fit1).all()
    assert (monitor.get_latest_solution() == -pop1).all()


@pytest.mark.parametrize("fitness_only", [True, False])
def test_pop_monitor(fitness_only):
    monitor = PopMonitor(fitness_only=fitness_only)
    algorithm = algorithms.CSO(lb=torch.zeros(5), ub=torch.ones(5), pop_size=4)
    workflow = workflows.StdWorkflow(algorithm, problems.numerical.Sphere(), monitors=[monitor])
    state = workflow.init(ex.PRNGKey(0))
    state = workflow.step(state)
    algo = state.get_child_state("algorithm")
    assert (monitor.get_latest_fitness() == algo.fitness).all()
    if not fitness_only:
        assert (monitor.get_latest_population() == algo.population).all()
    fig = monitor.plot()
    assert fig is not None and len(fig.data) >= 1


def test_evoxvis_monitor_roundtrip(tmp_path):
    import numpy as np
    import pyarrow as pa
    from evox_amd.monitors import EvoXVisMonitor
    monitor = EvoXVisMonitor("run", out_dir=str(tmp_path), batch_size=4)
    algorithm = algorithms.PSO(lb=torch.zeros(3), ub=torch.ones(3), pop_size=8)
    workflow = workflows.StdWorkflow(algorithm, problems.numerical.Sphere(), monitors=[monitor])
    state = workflow.init(ex.PRNGKey(0))
    for _ in range(10):
        state = workflow.step(state)
    monitor.close()
    table = pa.ipc.open_file(monitor.path).read_all()
    assert table.num_rows == 10
    assert table.column("generation").to_pylist() == list(range(10))
    meta = {k.decode(): v.decode() for k, v in table.schema.metadata.items()}
    assert meta["population_size"] == "8" and meta["fitness_dtype"] == "float32"
    fit = np.frombuffer(table.column("fitness")[0].as_py(), dtype=np.float32)
    assert fit.shape == (8,)
    pop = np.frombuffer(table.column("population")[0].as_py(), dtype=meta["population_dtype"]).reshape(8, 3)
    np.testing.assert_allclose((pop ** 2).sum(1), fit, rtol=1e-5)
    assert table.column("duration").to_pylist() == sorted(table.column("duration").to_pylist())


def test_plots_build():
    from evox_amd.vis_tools import plot
    hist1 = [torch.rand(10) for _ in range(4)]
    assert len(plot.plot_obj_space_1d(hist1).frames) == 4
    assert len(plot.plot_obj_space_1d(hist1, animation=False).data) == 4
    hist2 = [torch.rand(10, 2) for _ in range(3)]
    fig = plot.plot_obj_space_2d(hist2, problem_pf=torch.rand(20, 2), sort_points=True)
    assert len(fig.frames) == 3 and len(fig.data) == 2
    hist3 = [torch.rand(10, 3) for _ in range(3)]
    assert len(plot.plot_obj_space_3d(hist3, problem_pf=torch.rand(20, 3)).frames) == 3
    assert len(plot.plot_dec_space([torch.rand(10, 2) for _ in range(5)]).frames) == 5
