"""Numerical problems: known optima (scenarios of the reference's
``tests/test_classic_problems.py:8-60``), structural invariants of the ZDT / DTLZ / MaF / LSMOP
suites and the CEC 2022 suite.  The reference's MaF / LSMOP tests pin values computed from
``jax.random`` draws, which cannot be regenerated here (parity unpinned); the checks below pin
the mathematical properties instead: Pareto-optimal inputs land on the sampled front."""
import math

import pytest
import torch

import evox_amd as ex
from evox_amd import metrics, problems
from evox_amd.problems.numerical import CEC2022TestSuit, lsmop, maf

num = problems.numerical


@pytest.mark.parametrize("name,x,atol", [("Ackley", 0.0, 1e-5), ("Griewank", 0.0, 1e-6), ("Rastrigin", 0.0, 1e-6),
                                         ("Rosenbrock", 1.0, 1e-6), ("Schwefel", 420.9687462275036, 1e-3),
                                         ("Sphere", 0.0, 1e-6)])
def test_single_objective_optima(name, x, atol):
    prob = getattr(num, name)()
    state = prob.init(ex.PRNGKey(12345))
    F, state = prob.evaluate(state, torch.full((16, 2), x))
    torch.testing.assert_close(F, torch.zeros(16), atol=atol, rtol=0)


def test_dtlz1_front_sum():
    prob = num.DTLZ1(d=None, m=4)
    state = prob.init(ex.PRNGKey(0))
    F, _ = prob.evaluate(state, torch.full((16, 7), 0.5))
    torch.testing.assert_close(F.sum(1), torch.full((16,), 0.5))
    assert prob.pf().shape[1] == 4


@pytest.mark.parametrize("k", [2, 3, 4, 5, 6])
def test_dtlz_sphere_fronts(k):
    prob = getattr(num, f"DTLZ{k}")(m=3)
    state = prob.init(ex.PRNGKey(0))
    x = torch.rand(64, prob.d)
    x[:, 2:] = 0.0 if k == 6 else 0.5
    F, _ = prob.evaluate(state, x)
    torch.testing.assert_close((F ** 2).sum(1), torch.ones(64), atol=1e-4, rtol=0)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 6])
def test_zdt_optimal_front(k):
    prob = getattr(num, f"ZDT{k}")(n=10)
    state = prob.init(ex.PRNGKey(0))
    x = torch.zeros(50, 10)
    x[:, 0] = torch.linspace(0, 1, 50)
    F, _ = prob.evaluate(state, x)
    assert F.shape == (50, 2)
    from evox_amd.operators import non_dominated_sort
    F = F[non_dominated_sort(F) == 0]          # ZDT3: only parts of the g = 1 curve are optimal
    assert float(metrics.GD(prob.pf())(F)) < 0.02


def _maf_optimal_inputs(k, prob, n=512):
    d = prob.d
    x = torch.rand(n, d)
    if k in (1, 2, 3, 4, 5, 6):
        x[:, 2:] = 0.5
    elif k in (10, 12):
        scale = torch.arange(2, 2 * d + 1, 2, dtype=torch.float32)
        x = x * scale
        x[:, 2:] = 0.35 * scale[2:]
    elif k == 13:
        j = torch.arange(1, d + 1)
        x[:, 2:] = (2 * x[:, 1:2] * torch.sin(2 * math.pi * x[:, :1] + j * math.pi / d))[:, 2:]
    return x


@pytest.mark.parametrize("k,tol", [(1, 0.05), (2, 0.08), (3, 0.05), (4, 0.3), (5, 0.05), (6, 0.02), (10, 0.25), (12, 0.3),
                                   (13, 0.08)])
def test_maf_optimal_inputs_hit_front(k, tol):
    prob = getattr(maf, f"MaF{k}")(m=3, ref_num=100)
    state = prob.init(ex.PRNGKey(0))
    F, _ = prob.evaluate(state, _maf_optimal_inputs(k, prob))
    assert torch.isfinite(F).all()
    assert float(metrics.GD(prob.pf())(F)) < tol
    worse, _ = prob.evaluate(state, torch.rand(512, prob.d) * (1 if k not in (10, 12) else 2))
    assert float(metrics.GD(prob.pf())(worse)) > float(metrics.GD(prob.pf())(F))


@pytest.mark.parametrize("k", range(1, 16))
def test_maf_shapes(k):
    for m in (3, 5):
        prob = getattr(maf, f"MaF{k}")(m=m, ref_num=20)
        state = prob.init(ex.PRNGKey(0))
        F, _ = prob.evaluate(state, torch.rand(10, prob.d))
        assert F.shape == (10, m) and torch.isfinite(F).all()
        pf = prob.pf()
        assert pf.shape[1] == m and pf.shape[0] > 0 and torch.isfinite(pf).all()


def test_maf_polygon_helpers():
    assert not bool(maf.inside(8.5, 1.0, 0.0)) and bool(maf.inside(0.5, 1.0, 0.0))
    assert not bool(maf.inside(1.0, 1.0, 0.0)) and bool(maf.inside(0.0, 1.0, 0.0))
    p = torch.tensor([0.0, 0.0])
    t = torch.tensor
    assert not bool(maf.ray_intersect_segment(p, t([1.0, 1.0]), t([1.0, 2.0])))
    assert bool(maf.ray_intersect_segment(p, t([1.0, 1.0]), t([-1.0, -1.0])))
    assert bool(maf.ray_intersect_segment(p, t([1.0, 1.0]), t([1.0, -1.0])))
    assert not bool(maf.ray_intersect_segment(p, t([1.0, 0.0]), t([1.0, -1.0])))
    assert bool(maf.ray_intersect_segment(p, t([1.0, 0.0]), t([1.0, 1.0])))
    poly = t([[0, 1.0], [-0.5, -1], [0.5, -1]])
    assert bool(maf.point_in_polygon(poly, t([0.0, 0.0]))) and not bool(maf.point_in_polygon(poly, t([1.0, -1.0])))
    assert bool(maf.point_in_polygon(poly, t([0.0, 1.0]))) and not bool(maf.point_in_polygon(poly, t([-1.0, 1.0])))


def test_maf8_inside_polygon_is_nondominated():
    prob = maf.MaF8(m=3)
    state = prob.init(ex.PRNGKey(0))
    inside_pts = torch.tensor([[0.0, 0.0], [0.1, -0.2]])
    outside = torch.tensor([[3.0, 3.0]])
    F, _ = prob.evaluate(state, torch.cat([inside_pts, outside]))
    assert (F[2] > F[0]).all()


@pytest.mark.parametrize("k", range(1, 10))
def test_lsmop(k):
    prob = getattr(lsmop, f"LSMOP{k}")(m=3, d=60)
    state = prob.init(ex.PRNGKey(0))
    x = torch.rand(100, 60)
    F, _ = prob.evaluate(state, x)
    assert F.shape == (100, 3) and torch.isfinite(F).all()
    # solutions obeying the variable linkage have zero-centred distance variables
    idx = torch.arange(3, 61) / 60
    w = 1 + (torch.cos(idx * math.pi / 2) if prob.nonlinear_linkage else idx)
    xo = x.clone()
    xo[:, 2:] = x[:, :1] * 10 / w
    Fo, _ = prob.evaluate(state, xo)
    if k in (1, 2, 4):
        torch.testing.assert_close(Fo.sum(1), torch.ones(100), atol=1e-3, rtol=0)
    if k in (5, 8):
        torch.testing.assert_close((Fo ** 2).sum(1), torch.ones(100), atol=1e-3, rtol=0)
    assert prob.pf().shape[1] == 3


@pytest.mark.parametrize("f", range(1, 13))
def test_cec2022(f):
    prob = CEC2022TestSuit.create(f)
    state = prob.init()
    for d in (10, 20):
        opt = CEC2022TestSuit.optimum(f, d, torch.float64)[None, :]
        x = torch.cat([opt, torch.rand(8, d, dtype=torch.float64) * 200 - 100])
        y, _ = prob.evaluate(state, x)
        assert y.shape == (9,) and torch.isfinite(y).all()
        assert float(y[0]) < 1e-6, (f, d, float(y[0]))
        assert (y[1:] > 1.0).all()
    y32, _ = prob.evaluate(state, torch.rand(4, 10) * 200 - 100)
    assert y32.dtype == torch.float32
