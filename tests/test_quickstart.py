"""README quick-start of the reference (PSO pop=100 on 2-D Ackley through StdWorkflow)."""
import pytest
import torch

import evox_amd as ex
from evox_amd import algorithms, monitors, problems, workflows


def run(device, steps=100):
    pso = algorithms.PSO(lb=torch.full((2,), -32.0, device=device), ub=torch.full((2,), 32.0, device=device),
                         pop_size=100)
    monitor = monitors.EvalMonitor()
    wf = workflows.StdWorkflow(pso, problems.numerical.Ackley(), monitors=[monitor])
    state = wf.init(ex.PRNGKey(42))
    for _ in range(steps):
        state = wf.step(state)
    return monitor, state


def test_quickstart_cpu():
    monitor, state = run(torch.device("cpu"))
    assert float(monitor.get_best_fitness()) < 1e-2
    assert state.generation == 100
    assert len(monitor.get_history()) == 100


@pytest.mark.gpu
def test_quickstart_gpu_matches_cpu_trajectory(cuda):
    m_gpu, s_gpu = run(cuda, steps=20)
    m_cpu, s_cpu = run(torch.device("cpu"), steps=20)
    assert float(m_gpu.get_best_fitness()) < 1.0
    a = s_gpu.get_child_state("algorithm")
    b = s_cpu.get_child_state("algorithm")
    # same Philox stream on both sides: trajectories agree up to transcendental rounding
    assert torch.allclose(a.population.cpu(), b.population, atol=1e-2, rtol=1e-2)


def test_examples_run():
    """The scripts under examples/ are part of the documentation: they must run end to end."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    for name in ("pso_ackley.py", "cmaes_multi_gpu.py", "island_shade.py", "clustered_shade.py"):       # multi-process ones as single processes
        out = subprocess.run([sys.executable, os.path.join(root, "examples", name)], env=env, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "best fitness" in out.stdout


def test_graph_step_auto_without_a_gpu_is_eager():
    import torch
    import evox_amd as ex
    from evox_amd import algorithms, problems, workflows

    wf = workflows.StdWorkflow(algorithms.PSO(lb=torch.full((5,), -3.0), ub=torch.full((5,), 3.0), pop_size=20),
                               problems.numerical.Sphere(), graph_step="auto")
    ref = workflows.StdWorkflow(algorithms.PSO(lb=torch.full((5,), -3.0), ub=torch.full((5,), 3.0), pop_size=20),
                                problems.numerical.Sphere())
    s, r = wf.init(ex.PRNGKey(1)), ref.init(ex.PRNGKey(1))
    for _ in range(6):
        s, r = wf.step(s), ref.step(r)
    assert wf.graph_step is False and wf.graph_choice["mode"] == "eager"
    assert torch.equal(s.get_child_state("algorithm").population, r.get_child_state("algorithm").population)
